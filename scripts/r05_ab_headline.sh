#!/bin/bash
# the product library against variant libraries (scripts/build_variant.sh) on the headline WITH its host-to-host leg; same box, alternating processes.
# usage: scripts/r05_ab_headline.sh <tag> [<tag> ...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05ab
for rep in 1 2; do
for v in "" "$@"; do
  lib=$PWD/kmerutils_amd/libkmu${v:+_$v}.so
  KMU_LIB=$lib timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs > gpurun_out/r05ab/a$v.json 2> gpurun_out/r05ab/a$v.err || { echo "run $v failed"; tail -3 gpurun_out/r05ab/a$v.err; continue; }
  python3 - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05ab/a%s.json'%sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels']
print('variant %-8s: device leg %.2f ms  l1 %.2f  l2 %.2f  build %.2f | host to host %.2f ms %s | checks %s %s %s' % (sys.argv[1] or 'product', d['ms_per_step'], k['k_part_scatter1']['avg_ms'], k['k_arr_scatter']['avg_ms'], k['k_part_build_q']['avg_ms'], d["host_to_host"]["ms_per_step"], {k: round(v, 1) for k, v in d["host_to_host"].get("kernel_ms_per_step", {}).items() if v >= 0.5}, d["checks"].get("parity_rows_ok"), d['checks'].get('parity_counts_ok'), d['checks'].get('host_leg_equals_device_leg')))
PY
done
done
