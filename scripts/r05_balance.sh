#!/bin/bash
# region build with / without the region-bal requests (KMU_BUILD_BALANCE), same box, alternating processes; count-only headline
mkdir -p gpurun_out/r05bal
for v in 1 0 1 0; do
  KMU_LIB=$PWD/kmerutils_amd/libkmu_bal$v.so timeout -k 10 120 python bench.py --workload ont_k31_count --steps 4 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/r05bal/a$v.json 2> gpurun_out/r05bal/a$v.err || { echo "run $v failed"; tail -3 gpurun_out/r05bal/a$v.err; continue; }
  python3 - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05bal/a%s.json'%sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels']
print('bal=%s: step %.2f ms  l1 %.2f  l2 %.2f  build %.2f' % (sys.argv[1], d['ms_per_step'], k['k_part_scatter1']['avg_ms'], k['k_arr_scatter']['avg_ms'], k['k_part_build_q']['avg_ms']))
PY
done
