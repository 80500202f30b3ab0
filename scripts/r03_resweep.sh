#!/bin/bash
# the set / unit counts of the shared segments once more, with the step table in place
cd $GRAFT_REPO_ROOT
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload ont_k31_count --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-host-leg > gpurun_out/rs_$label.json 2> gpurun_out/rs_$label.err
  rc=$?
  grep -q "Memory access fault" gpurun_out/rs_$label.err && { echo "GPU FAULT in $label"; exit 1; }
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/rs_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/rs_$label.json').read().strip().splitlines()[-1])
print('$label', 'dev ms',round(d['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if '+' not in k})"
}
run s16_c16 KMU_X=1
run s8_c16 KMU_COUNT_SEG_SHARED=8
run s32_c16 KMU_COUNT_SEG_SHARED=32
run s16_c32 KMU_COUNT_L2_SHARED=32
run s16_c8 KMU_COUNT_L2_SHARED=8
run s32_c32 KMU_COUNT_SEG_SHARED=32 KMU_COUNT_L2_SHARED=32
run s16_c16b KMU_X=1
