#!/bin/bash
# level-1 units that persist across the chunk rounds of the host leg: tests first, then the A/B (alternating processes)
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_count_quot.py -x -q -m gpu -k "sketch_count or pipeline" > gpurun_out/t_rounds.log 2>&1
rc=$?
tail -5 gpurun_out/t_rounds.log
[ $rc -eq 0 ] || exit 1
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/ab_rounds_$label.json 2> gpurun_out/ab_rounds_$label.err
  rc=$?
  if grep -q "Memory access fault" gpurun_out/ab_rounds_$label.err; then echo "GPU FAULT in $label"; exit 1; fi
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_rounds_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_rounds_$label.json').read().strip().splitlines()[-1])
print('$label', 'dev ms',round(d['ms_per_step'],2),'host ms', round(d['host_to_host']['ms_per_step'],2), d['checks'].get('host_leg_equals_device_leg'))"
}
run rounds1 KMU_X=1
run rounds0 KMU_COUNT_SEG_ROUNDS=0
run rounds1b KMU_X=1
run rounds0b KMU_COUNT_SEG_ROUNDS=0
run rounds1_min16 KMU_COUNT_SEG_ROUND_MIN=16
