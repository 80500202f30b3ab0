#!/bin/bash
# level 1 with segments shared by the workgroups of an XCD (cursors) against a segment per unit: tests, then the A/B
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_count_quot.py tests/test_gpu_parity.py -x -q -m gpu -k "count or pipeline or single_pass" > gpurun_out/t_shared.log 2>&1
rc=$?
tail -5 gpurun_out/t_shared.log
[ $rc -eq 0 ] || exit 1
fi
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity ${EXTRA:-} > gpurun_out/ab_shared_$label.json 2> gpurun_out/ab_shared_$label.err
  rc=$?
  if grep -q "Memory access fault" gpurun_out/ab_shared_$label.err; then echo "GPU FAULT in $label"; exit 1; fi
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_shared_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_shared_$label.json').read().strip().splitlines()[-1])
h=d.get('host_to_host') or {}
print('$label', 'dev ms',round(d['ms_per_step'],2),'host ms', round(h.get('ms_per_step',0),2), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if k.startswith('k_part') or k.startswith('k_arr')}, d['checks'])"
}
export EXTRA=--no-host-leg
run sh16 KMU_COUNT_SEG_SHARED=16
run sh32 KMU_COUNT_SEG_SHARED=32
run sh64 KMU_COUNT_SEG_SHARED=64
run sh2 KMU_COUNT_SEG_SHARED=2
run sh16div KMU_COUNT_SEG_SHARED=16 KMU_COUNT_SEG_SETMAP=div
run sh8div KMU_COUNT_SEG_SHARED=8 KMU_COUNT_SEG_SETMAP=div
run sh16b KMU_COUNT_SEG_SHARED=16
run sh32b KMU_COUNT_SEG_SHARED=32
run sh128 KMU_COUNT_SEG_SHARED=128
run sh1 KMU_COUNT_SEG_SHARED=1
