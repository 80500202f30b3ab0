#!/bin/bash
# regions of 8 192 slots (KMU_COUNT_RBITS=13): count tests on tables that take them, then the count bench A/B; stops at the first failure
cd $GRAFT_REPO_ROOT
KMU_COUNT_RBITS=13 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -q -x -k "single_pass or two_level or sketch_count or full_size" > gpurun_out/t_r03e.log 2>&1; rc=$?; echo rc=$rc >> gpurun_out/t_r03e.log; tail -4 gpurun_out/t_r03e.log
grep -q "Memory access fault" gpurun_out/t_r03e.log && { echo "GPU FAULT in the tests"; exit 1; }
[ $rc -eq 0 ] || exit 1
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload ont_k31_count --steps 4 --warmup 1 --no-cpu-baseline --no-host-leg > gpurun_out/ab_rb_$label.json 2> gpurun_out/ab_rb_$label.err
  rc=$?
  if grep -q "Memory access fault" gpurun_out/ab_rb_$label.err; then echo "GPU FAULT in $label"; exit 1; fi
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_rb_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_rb_$label.json').read().strip().splitlines()[-1])
print('$label', 'ms',round(d['ms_per_step'],2),{k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if '+' not in k}, d['checks'])"
}
for i in 1 2 3; do
run rb13_$i KMU_COUNT_RBITS=13
run rb12_$i KMU_X=1
done
