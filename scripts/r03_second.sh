#!/bin/bash
# round 3, second GPU session: the thread-rank tests, sketch parity with the new uq kernel, A/B of the sketch, full suite
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_comm.py -q -x -k "eight or rounds" > gpurun_out/t_r03b.log 2>&1; echo rc=$? >> gpurun_out/t_r03b.log; tail -4 gpurun_out/t_r03b.log
grep -q "^rc=0" gpurun_out/t_r03b.log || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_pipeline.py -q -x -k "probminhash or two_kernel or many_reads or sketch or full_size or ranges" > gpurun_out/t_r03c.log 2>&1; echo rc=$? >> gpurun_out/t_r03c.log; tail -4 gpurun_out/t_r03c.log
grep -q "^rc=0" gpurun_out/t_r03c.log || exit 1
for v in 1 0 1 0; do
  KMU_PMH_UQTAB=$v timeout -k 10 200 python bench.py --workload ont_k31_sketch --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg > gpurun_out/ab_uqtab_$v.json 2> gpurun_out/ab_uqtab_$v.err || { echo bench failed; tail -5 gpurun_out/ab_uqtab_$v.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_uqtab_$v.json').read().strip().splitlines()[-1])
print('UQTAB=$v ms',round(d['ms_per_step'],2),{k:round(v['avg_ms'],2) for k,v in d['kernels'].items()},d['checks'])"
done
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err || { echo bench failed; tail -20 gpurun_out/bench_r03a.err; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/bench_r03a.json').read().strip().splitlines()[-1])
print('value',d['value'],'h2h',d['value_host_to_host'],'ms',d['ms_per_step']);print(json.dumps(d['roofline'],indent=0));print({k:round(v['avg_ms'],2) for k,v in d['kernels'].items()});print(d['checks']);print(d['alu'])"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_r03_full.log 2>&1; echo rc=$? >> gpurun_out/t_r03_full.log; tail -5 gpurun_out/t_r03_full.log
