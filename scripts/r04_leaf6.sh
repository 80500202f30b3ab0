#!/bin/bash
# leaf items of 6 bytes (KMU_COUNT_LEAF6=1) against 8: the count-only benches, same box, alternating
cd $GRAFT_REPO_ROOT
for w in ont_k31_count c4_count; do
  for rep in 1 2; do
    for l6 in 0 1; do
      KMU_COUNT_LEAF6=$l6 timeout -k 10 200 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg 2> gpurun_out/l6.err | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w leaf6=$l6', round(d['ms_per_step'],2), {k: round(v['avg_ms'],2) for k,v in d['kernels'].items() if '+' not in k}, d['checks'])" || { tail -5 gpurun_out/l6.err; exit 1; }
    done
  done
done
