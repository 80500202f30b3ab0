#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_count_quot.py -q -x -k "sketch_count" > gpurun_out/t_r03j.log 2>&1; rc=$?; echo rc=$rc >> gpurun_out/t_r03j.log; tail -4 gpurun_out/t_r03j.log
grep -q "Memory access fault" gpurun_out/t_r03j.log && { echo "GPU FAULT in the tests"; exit 1; }
[ $rc -eq 0 ] || exit 1
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/ab_host_$label.json 2> gpurun_out/ab_host_$label.err
  rc=$?
  if grep -q "Memory access fault" gpurun_out/ab_host_$label.err; then echo "GPU FAULT in $label"; exit 1; fi
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_host_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_host_$label.json').read().strip().splitlines()[-1])
print('$label', 'dev ms',round(d['ms_per_step'],2),'host ms', round(d['host_to_host']['ms_per_step'],2), 'scatter1', round(d['kernels']['k_part_scatter1']['avg_ms'],2), d['checks'].get('host_leg_equals_device_leg'))"
}
run two_a KMU_X=1
run one_a KMU_PIPE_TWO=0
run two_b KMU_X=1
run one_b KMU_PIPE_TWO=0
run two_c256 KMU_PIPE_CHUNK_MB=256
run two_c384 KMU_PIPE_CHUNK_MB=384
