#!/bin/bash
# level 2 with the leaves of a bin shared by several units (cursors): tests with it forced, then the A/B
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
KMU_COUNT_L2_THREADS=512 timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_count_quot.py tests/test_gpu_parity.py -x -q -m gpu -k "count or pipeline or single_pass" > gpurun_out/t_shared2.log 2>&1
rc=$?
tail -5 gpurun_out/t_shared2.log
[ $rc -eq 0 ] || exit 1
fi
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-host-leg > gpurun_out/ab_shared2_$label.json 2> gpurun_out/ab_shared2_$label.err
  rc=$?
  if grep -q "Memory access fault" gpurun_out/ab_shared2_$label.err; then echo "GPU FAULT in $label"; exit 1; fi
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_shared2_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_shared2_$label.json').read().strip().splitlines()[-1])
print('$label', 'dev ms',round(d['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if k.startswith('k_part') or k.startswith('k_arr')}, d['checks'])"
}
run t1024 KMU_X=1
run t512 KMU_COUNT_L2_THREADS=512
run t512_c8 KMU_COUNT_L2_THREADS=512 KMU_COUNT_L2_SHARED=8
run t512_c32 KMU_COUNT_L2_THREADS=512 KMU_COUNT_L2_SHARED=32
run t1024b KMU_X=1
run t512b KMU_COUNT_L2_THREADS=512
run t512_c64 KMU_COUNT_L2_THREADS=512 KMU_COUNT_L2_SHARED=64
