#!/bin/bash
# k_pmh_points: cheap test without the 1 / w look-up inside the weight-1 prefix (product) against with it (variant nounit)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden_signatures.py -x -q -m gpu -k "probminhash or points or two_kernel or golden" > gpurun_out/t_unitw.log 2>&1
rc=$?; tail -2 gpurun_out/t_unitw.log; [ $rc -eq 0 ] || exit 1
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload ont_k31_sketch --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-host-leg > gpurun_out/ab_uw_$label.json 2> gpurun_out/ab_uw_$label.err
  rc=$?
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_uw_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_uw_$label.json').read().strip().splitlines()[-1])
print('$label', 'dev ms',round(d['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if '+' not in k}, d['checks']['sig_checksum'])"
}
V=$GRAFT_REPO_ROOT/kmerutils_amd/libkmu_nounit.so
run unit KMU_X=1
run nounit KMU_LIB=$V
run unit_b KMU_X=1
run nounit_b KMU_LIB=$V
run unit_c KMU_X=1
