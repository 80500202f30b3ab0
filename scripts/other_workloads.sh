#!/bin/bash
# the non-headline BASELINE.json configurations that fit one GPU (parity-test cases; measured for DESIGN.md only)
cd $GRAFT_REPO_ROOT
for w in ${WORKLOADS:-c3_k8 c2_count c1_super c5_aa}; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 > gpurun_out/wl_$w.json 2> gpurun_out/wl_$w.err || echo "FAILED $w"
  python3 -c "
import json
d=json.loads(open('gpurun_out/wl_$w.json').read().strip().splitlines()[-1])
print('$w', round(d['value'],4), d['unit'], 'ms/step', round(d['ms_per_step'],3), 'cpu', d.get('cpu_baseline',{}).get('value'))
for k,v in d['kernels'].items(): print('   ', k, round(v['avg_ms'],3))
"
done
