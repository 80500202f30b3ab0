#!/bin/bash
# regions of 2048 slots (256-thread builds, eight workgroups per CU) against 4096 (512 threads, four); count-only headline, same box
mkdir -p gpurun_out/r05rb
for v in 12 11 12 11; do
  KMU_LIB=$PWD/kmerutils_amd/libkmu_rb$v.so timeout -k 10 120 python bench.py --workload ont_k31_count --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r05rb/a$v.json 2> gpurun_out/r05rb/a$v.err || { echo "run $v failed"; tail -3 gpurun_out/r05rb/a$v.err; continue; }
  python3 - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05rb/a%s.json'%sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels']
print('region bits %s: step %.2f ms  l1 %.2f  l2 %.2f  build %.2f  %s' % (sys.argv[1], d['ms_per_step'], k['k_part_scatter1']['avg_ms'], k['k_arr_scatter']['avg_ms'], k['k_part_build_q']['avg_ms'], {a: b for a, b in d['checks'].items() if 'ok' in a}))
PY
done
