#!/bin/bash
# the product library against variant libraries (scripts/build_variant.sh <tag> ...: kmerutils_amd/libkmu_<tag>.so), same box, alternating processes;
# count-only headline (ont_k31_count), ms per kernel.  usage: scripts/r05_rbits.sh <tag> [<tag> ...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05rb
for rep in 1 2; do
for v in "" "$@"; do
  lib=$PWD/kmerutils_amd/libkmu${v:+_$v}.so
  KMU_LIB=$lib timeout -k 10 120 python bench.py --workload ont_k31_count --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r05rb/a$v.json 2> gpurun_out/r05rb/a$v.err || { echo "run $v failed"; tail -3 gpurun_out/r05rb/a$v.err; continue; }
  python3 - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05rb/a%s.json'%sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels']
print('variant %-8s: step %.2f ms  l1 %.2f  l2 %.2f  build %.2f  parity %s' % (sys.argv[1] or 'product', d['ms_per_step'], k['k_part_scatter1']['avg_ms'], k['k_arr_scatter']['avg_ms'], k['k_part_build_q']['avg_ms'], d['checks'].get('parity_counts_ok')))
PY
done
done
