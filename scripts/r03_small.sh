#!/bin/bash
# kernel times of a ninth of the bench batch, device resident (what a chunk of the host leg costs without the copies around it)
cd $GRAFT_REPO_ROOT
for n in 83000 166000 746333; do
  timeout -k 10 300 python bench.py --reads $n --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-host-leg > gpurun_out/small_$n.json 2> gpurun_out/small_$n.err || { tail -3 gpurun_out/small_$n.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/small_$n.json').read().strip().splitlines()[-1])
print($n, 'ms',round(d['ms_per_step'],2), {k:(round(v['avg_ms'],3), v.get('launches_per_step')) for k,v in d['kernels'].items() if '+' not in k})"
done
