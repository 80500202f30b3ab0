#!/bin/bash
# The exchange of a distributed count UNDER the sketch kernels, on one GPU (KMU_BENCH_FORCE_COMM=1: the N-rank code path with a
# communicator of one rank): how long the all-to-all of the rank's own 3.3 GB of records takes while the persistent sketch kernels
# hold the CUs, by KMU_PMH_RESERVE_CUS (workgroups the sketch kernels leave out) and by how the share travels (device copy = the
# default for a rank's own share, or through RCCL like a peer's: KMU_COMM_SELF_RCCL=1).  Output: gpurun_out/r05_reserve.txt
mkdir -p gpurun_out/r05r
out=gpurun_out/r05_reserve.txt; : > $out
for self in 0 1; do
  for R in ${RESERVES:-0 8 16 32 64}; do
    KMU_BENCH_FORCE_COMM=1 KMU_PMH_RESERVE_CUS=$R KMU_COMM_SELF_RCCL=$self timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-host-leg --no-parity \
      > gpurun_out/r05r/res_${self}_$R.json 2> gpurun_out/r05r/res_${self}_$R.err || { echo "reserve $R self_rccl $self failed" | tee -a $out; continue; }
    python3 - $self $R >> $out <<'PY'
import json,sys
self_,R=sys.argv[1:3]
d=json.loads(open('gpurun_out/r05r/res_%s_%s.json'%(self_,R)).read().strip().splitlines()[-1])
k=d['kernels']; c=d['comm']
sk=[n for n in k if n.startswith('k_multiset_uq+')]
print('self share via %-11s reserve %3s CUs: step %.2f ms  sketch unit %.2f ms  exchange_ms %.2f (per step, %d exchanges)  records %.2f GB' % (
    'RCCL' if self_=='1' else 'device copy', R, d['ms_per_step'], k[sk[0]]['avg_ms'] if sk else -1, c['exchange_ms']/max(1,c['exchanges']), c['exchanges'], c['records_local']*12/1e9))
PY
  done
done
# the same exchange without a sketch kernel in its way
KMU_BENCH_FORCE_COMM=1 timeout -k 10 150 python bench.py --workload ont_k31_count --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-parity > gpurun_out/r05r/res_count.json 2> gpurun_out/r05r/res_count.err && python3 - >> $out <<'PY'
import json
d=json.loads(open('gpurun_out/r05r/res_count.json').read().strip().splitlines()[-1]); c=d['comm']
print('count only (no sketch kernels), device copy: step %.2f ms  exchange_ms %.2f' % (d['ms_per_step'], c['exchange_ms']/max(1,c['exchanges'])))
PY
cat $out
