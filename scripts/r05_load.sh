#!/bin/bash
# region build vs table load factor (KMU_COUNT_LOAD, per cent), count-only headline workload; same box, one process per setting
mkdir -p gpurun_out/r05b
for L in ${LOADS:-47 57 70 80}; do
  KMU_COUNT_LOAD=$L timeout -k 10 120 python bench.py --workload ont_k31_count --steps 3 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/r05b/load_$L.json 2> gpurun_out/r05b/load_$L.err || exit 1
  python3 - $L <<'PY'
import json,sys
L=sys.argv[1]
d=json.loads(open('gpurun_out/r05b/load_%s.json'%L).read().strip().splitlines()[-1])
k=d['kernels']
print('load %s: step %.2f ms  l1 %.2f  l2 %.2f  build %.2f  table %.1f GB' % (L, d['ms_per_step'], k['k_part_scatter1']['avg_ms'], k['k_arr_scatter']['avg_ms'], k['k_part_build_q']['avg_ms'], (k['k_part_build_q']['design_bytes']-d['config']['kmers_per_gpu']*8)/1e9))
PY
done
