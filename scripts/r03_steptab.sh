#!/bin/bash
# level 1 with the reads of the wave steps from a table made once per call (product) against looked up per step (variant notab)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_count_quot.py tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py -x -q -m gpu -k "count or pipeline or single_pass or sweep" > gpurun_out/t_steptab.log 2>&1
rc=$?; tail -2 gpurun_out/t_steptab.log
grep -q "Memory access fault" gpurun_out/t_steptab.log && { echo GPU FAULT; exit 1; }
[ $rc -eq 0 ] || exit 1
VARIANT=${VARIANT:-notab} bash scripts/r03_prerank.sh
