#!/bin/bash
# full -m gpu suite + default bench line; stops at the first failure (and at any GPU fault)
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/t_r03_full.log 2>&1; rc=$?; echo rc=$rc >> gpurun_out/t_r03_full.log; tail -5 gpurun_out/t_r03_full.log
grep -q "Memory access fault" gpurun_out/t_r03_full.log && { echo "GPU FAULT"; exit 1; }
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r03b.json 2> gpurun_out/bench_r03b.err || { echo bench failed; tail -20 gpurun_out/bench_r03b.err; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/bench_r03b.json').read().strip().splitlines()[-1])
print('value',d['value'],'h2h',d['value_host_to_host'],'ms',d['ms_per_step'], 'h2h ms', d['host_to_host']['ms_per_step']);print({k:round(v['avg_ms'],2) for k,v in d['kernels'].items()});print(d['checks']);print(d['alu']); print({k:(round(v['frac'],4), round(v['avg_launch_ms'],2)) for k,v in d['roofline']['units'].items()})"
