#!/bin/bash
# the launcher + N-rank parity on one GPU: 5 ranks share the card (the box allows six processes), both routes
cd $GRAFT_REPO_ROOT
for route in occurrences merge; do
  KMU_BENCH_BACKEND=gloo KMU_COUNT_ROUTE=$route timeout -k 10 500 python bench.py --gpus 5 --reads 30000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/rehearse_$route.json 2> gpurun_out/rehearse_$route.err
  rc=$?
  grep -q "Memory access fault" gpurun_out/rehearse_$route.err && { echo GPU FAULT; exit 1; }
  [ $rc -eq 0 ] || { echo "$route failed rc=$rc"; tail -8 gpurun_out/rehearse_$route.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/rehearse_$route.json').read().strip().splitlines()[-1])
print('$route', 'n_gpus', d['n_gpus'], 'ms', round(d['ms_per_step'],2), d['checks'])"
done
