#!/bin/bash
# The two routes of a distributed add timed on ONE GPU (RCCL communicator of one rank: the all-to-all goes to self), for the
# cost model's constants (kmu_count.hip route_model) and for DESIGN.md 4: per-kernel times and the exchange volumes both
# routes would move at N ranks.  Output: gpurun_out/routes_<workload>_<route>.json
cd $GRAFT_REPO_ROOT
export KMU_BENCH_FORCE_COMM=1 NCCL_SOCKET_IFNAME=${NCCL_SOCKET_IFNAME:-lo}
for w in ${ROUTE_WORKLOADS:-c4_count ont_k31_count}; do
  for r in occurrences merge auto; do
    if [ "$r" = auto ]; then unset KMU_COUNT_ROUTE; else export KMU_COUNT_ROUTE=$r; fi
    timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/routes_${w}_$r.json 2> gpurun_out/routes_${w}_$r.err || { echo "$w $r failed"; tail -5 gpurun_out/routes_${w}_$r.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('gpurun_out/routes_${w}_$r.json').read().strip().splitlines()[-1])
print('$w', '$r', 'ms', round(d['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items()}, d['comm'], d['checks'])"
  done
done
