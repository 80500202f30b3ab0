#!/bin/bash
# level 1: an item's rank taken as soon as its hash is made (variant library libkmu_pr.so) against the ranks of a tile in one loop
cd $GRAFT_REPO_ROOT
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload ont_k31_count --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg > gpurun_out/ab_pr_$label.json 2> gpurun_out/ab_pr_$label.err
  rc=$?
  if grep -q "Memory access fault" gpurun_out/ab_pr_$label.err; then echo "GPU FAULT in $label"; exit 1; fi
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_pr_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_pr_$label.json').read().strip().splitlines()[-1])
print('$label', 'dev ms',round(d['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if k.startswith('k_part') or k.startswith('k_arr')}, d['checks'])"
}
V=$GRAFT_REPO_ROOT/kmerutils_amd/libkmu_${VARIANT:-pr}.so
run var KMU_LIB=$V
run base KMU_X=1
run var_b KMU_LIB=$V
run base_b KMU_X=1
run var_c KMU_LIB=$V
