#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_pipeline.py -q -x -k "probminhash or two_kernel or many_reads or sketch or full_size or ranges" > gpurun_out/t_r03g.log 2>&1; rc=$?; echo rc=$rc >> gpurun_out/t_r03g.log; tail -4 gpurun_out/t_r03g.log
grep -q "Memory access fault" gpurun_out/t_r03g.log && { echo "GPU FAULT in the tests"; exit 1; }
[ $rc -eq 0 ] || exit 1
KMU_PMH_UQTAB=0 AB_LIBS="h old h old" bash scripts/r03_ab_sketch.sh
