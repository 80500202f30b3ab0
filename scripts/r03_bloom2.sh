#!/bin/bash
# k_multiset_uq with two bitmap positions per key (a unique key in the collision groups with probability ~(2n/2^bits)^2 instead of n/2^bits)
# against one: measured with this script (r03: 13.8 against 12.6 ms per launch, rows identical), the -DKMU_UQ_BLOOM2 code then removed
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py tests/test_golden_signatures.py -x -q -m gpu -k "probminhash or points or sketch or pmh or two_kernel or golden or smallk or short or many_reads" > gpurun_out/t_bloom2.log 2>&1
rc=$?; tail -2 gpurun_out/t_bloom2.log
grep -q "Memory access fault" gpurun_out/t_bloom2.log && { echo GPU FAULT; exit 1; }
[ $rc -eq 0 ] || exit 1
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload ont_k31_sketch --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg > gpurun_out/ab_b2_$label.json 2> gpurun_out/ab_b2_$label.err
  rc=$?
  grep -q "Memory access fault" gpurun_out/ab_b2_$label.err && { echo "GPU FAULT in $label"; exit 1; }
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_b2_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_b2_$label.json').read().strip().splitlines()[-1])
print('$label', 'dev ms',round(d['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if '+' not in k}, d['checks']['sig_checksum'], d['checks'].get('parity_rows_ok'))"
}
V=$GRAFT_REPO_ROOT/kmerutils_amd/libkmu_bl1.so
run two KMU_X=1
run one KMU_LIB=$V
run two_b KMU_X=1
run one_b KMU_LIB=$V
run two_c KMU_X=1
