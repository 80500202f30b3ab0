#!/bin/bash
# one-off: the fuzz sweeps at eight times their seeds (1 120 cases), once as they are and once with the round's new routes forced
cd $GRAFT_REPO_ROOT
KMU_FUZZ_SCALE=8 timeout -k 10 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider > gpurun_out/t_fuzz8a.log 2>&1; echo rc=$? >> gpurun_out/t_fuzz8a.log; tail -2 gpurun_out/t_fuzz8a.log
grep -q "Memory access fault" gpurun_out/t_fuzz8a.log && { echo GPU FAULT; exit 1; }
grep -q "^rc=0" gpurun_out/t_fuzz8a.log || exit 1
KMU_FUZZ_SCALE=8 KMU_COUNT_SEG=2 KMU_PMH_PTS_LONG=1024 KMU_PMH_SPLIT=1 timeout -k 10 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider > gpurun_out/t_fuzz8b.log 2>&1; echo rc=$? >> gpurun_out/t_fuzz8b.log; tail -2 gpurun_out/t_fuzz8b.log
grep -q "Memory access fault" gpurun_out/t_fuzz8b.log && { echo GPU FAULT; exit 1; }
