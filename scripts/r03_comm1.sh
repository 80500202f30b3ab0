#!/bin/bash
# the N-rank code path (communicator, census, route, all-to-all to self, finalize) on one rank: kernel times
cd $GRAFT_REPO_ROOT
for route in occurrences; do
  KMU_BENCH_FORCE_COMM=1 KMU_COUNT_ROUTE=$route timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-leg > gpurun_out/comm1_$route.json 2> gpurun_out/comm1_$route.err
  rc=$?
  grep -q "Memory access fault" gpurun_out/comm1_$route.err && { echo GPU FAULT; exit 1; }
  [ $rc -eq 0 ] || { tail -5 gpurun_out/comm1_$route.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/comm1_$route.json').read().strip().splitlines()[-1])
print('$route', round(d['ms_per_step'],2), {k:(round(v['avg_ms'],2), v.get('launches_per_step')) for k,v in d['kernels'].items() if '+' not in k}, d['checks'])"
done
