#!/bin/bash
# k-mer reverse complements as bit fields of the lane's reversed window (variant libkmu_rcw.so): count tests, then the A/B
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/kmerutils_amd/libkmu_rcw.so
KMU_LIB=$V timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_count_quot.py -x -q -m gpu -k "count" > gpurun_out/t_rcw.log 2>&1
rc=$?
tail -3 gpurun_out/t_rcw.log
[ $rc -eq 0 ] || exit 1
VARIANT=rcw bash scripts/r03_prerank.sh
