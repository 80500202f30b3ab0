#!/bin/bash
# round 3, first GPU session: VALU issue table, the new tests, the full -m gpu suite, the default bench line
cd $GRAFT_REPO_ROOT
scripts/micro/valu_issue > gpurun_out/r03_valu_issue.txt 2>&1; tail -12 gpurun_out/r03_valu_issue.txt
timeout -k 10 900 python -m pytest tests/test_gpu_comm.py -q -k "eight or rounds" > gpurun_out/t_r03a.log 2>&1; echo rc=$? >> gpurun_out/t_r03a.log; tail -15 gpurun_out/t_r03a.log
grep -q "^rc=0" gpurun_out/t_r03a.log || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err || { echo bench failed; tail -20 gpurun_out/bench_r03a.err; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/bench_r03a.json').read().strip().splitlines()[-1])
print('value',d['value'],'h2h',d['value_host_to_host'],'ms',d['ms_per_step']);print(json.dumps(d['roofline'],indent=0));print({k:round(v['avg_ms'],2) for k,v in d['kernels'].items()});print(d['checks'])"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_r03_full.log 2>&1; echo rc=$? >> gpurun_out/t_r03_full.log; tail -5 gpurun_out/t_r03_full.log
