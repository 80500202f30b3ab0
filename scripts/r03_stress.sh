#!/bin/bash
# the GPU suite with the round's new routes FORCED on every call: single-pass partition with shared segments on every count
# (KMU_COUNT_SEG=2), long-read points by workgroups from 1 024 list entries on.  Failures of tests that assert on the kernels a
# route launches are expected; rows / tables that differ are not.
cd $GRAFT_REPO_ROOT
KMU_COUNT_SEG=2 KMU_PMH_PTS_LONG=1024 timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/t_stress.log 2>&1
echo rc=$? >> gpurun_out/t_stress.log
grep -q "Memory access fault" gpurun_out/t_stress.log && { echo GPU FAULT; exit 1; }
tail -25 gpurun_out/t_stress.log
