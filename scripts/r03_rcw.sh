#!/bin/bash
# k-mer reverse complements as bit fields of the lane's reversed window: the -DKMU_RC_WINDOW=1 code was measured with this script (r03) and
# then removed from kmu_count.hip (no gain) -- the script stays as the record of the A/B; same for -DKMU_SCATTER_PRERANK (r03_prerank.sh)
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/kmerutils_amd/libkmu_rcw.so
KMU_LIB=$V timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_count_quot.py -x -q -m gpu -k "count" > gpurun_out/t_rcw.log 2>&1
rc=$?
tail -3 gpurun_out/t_rcw.log
[ $rc -eq 0 ] || exit 1
VARIANT=rcw bash scripts/r03_prerank.sh
