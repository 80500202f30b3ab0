#!/bin/bash
# from which batch size on the single-pass partition beats the exact levels (shared segments: margins 16 x smaller than round 2's)
cd $GRAFT_REPO_ROOT
for n in 100 200 500 1000; do
  for seg in default 2; do
    if [ $seg = default ]; then E=""; else E="KMU_COUNT_SEG=2"; fi
    env $E timeout -k 10 200 python bench.py --workload ont_k31_count --reads $n --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-host-leg > gpurun_out/segthr_${n}_$seg.json 2> gpurun_out/segthr_${n}_$seg.err || { tail -3 gpurun_out/segthr_${n}_$seg.err; exit 1; }
    python3 -c "
import json;d=json.loads(open('gpurun_out/segthr_${n}_$seg.json').read().strip().splitlines()[-1])
print($n, '$seg', 'Mbases', round(d['config'].get('bases_per_gpu',0)/1e6,1), 'ms', round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items() if '+' not in k})"
  done
done
