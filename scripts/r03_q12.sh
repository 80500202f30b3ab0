#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_count_quot.py tests/test_gpu_parity.py -q -x -k "quot or single_pass or two_level or count_parity or add_kmers" > gpurun_out/t_r03f.log 2>&1; rc=$?; echo rc=$rc >> gpurun_out/t_r03f.log; tail -4 gpurun_out/t_r03f.log
grep -q "Memory access fault" gpurun_out/t_r03f.log && { echo "GPU FAULT in the tests"; exit 1; }
[ $rc -eq 0 ] || exit 1
AB="q12a:KMU_X=1 plain_a:KMU_BUILD_ABLATE=64 q12b:KMU_X=1 plain_b:KMU_BUILD_ABLATE=64" bash scripts/r03_ab_env.sh
