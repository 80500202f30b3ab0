#!/bin/bash
# level 1 on 512-thread workgroups, two per CU (needs 1 024 level-1 bins: KMU_COUNT_B1=10): measured with this script (r03), the
# KMU_COUNT_L1_THREADS code was then removed from kmu_count.hip (15.7-16.2 ms against 14.4); the script stays as the record
cd $GRAFT_REPO_ROOT
KMU_COUNT_B1=10 KMU_COUNT_L1_THREADS=512 timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_count_quot.py tests/test_gpu_parity.py -x -q -m gpu -k "count or pipeline or single_pass" > gpurun_out/t_l1t.log 2>&1
rc=$?; tail -2 gpurun_out/t_l1t.log
grep -q "Memory access fault" gpurun_out/t_l1t.log && { echo GPU FAULT; exit 1; }
[ $rc -eq 0 ] || exit 1
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload ont_k31_count --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg > gpurun_out/ab_l1t_$label.json 2> gpurun_out/ab_l1t_$label.err
  rc=$?
  if grep -q "Memory access fault" gpurun_out/ab_l1t_$label.err; then echo "GPU FAULT in $label"; exit 1; fi
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_l1t_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_l1t_$label.json').read().strip().splitlines()[-1])
print('$label', 'dev ms',round(d['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if k.startswith('k_part') or k.startswith('k_arr')}, d['checks'].get('parity_counts_ok'))"
}
run base KMU_X=1
run t512 KMU_COUNT_B1=10 KMU_COUNT_L1_THREADS=512
run b10 KMU_COUNT_B1=10
run t512_s32 KMU_COUNT_B1=10 KMU_COUNT_L1_THREADS=512 KMU_COUNT_SEG_SHARED=32
run base_b KMU_X=1
run t512_b KMU_COUNT_B1=10 KMU_COUNT_L1_THREADS=512
run t512_s8 KMU_COUNT_B1=10 KMU_COUNT_L1_THREADS=512 KMU_COUNT_SEG_SHARED=8
