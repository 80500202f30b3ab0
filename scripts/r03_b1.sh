#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload ont_k31_count --steps 4 --warmup 1 --no-cpu-baseline --no-host-leg > gpurun_out/ab_b1_$label.json 2> gpurun_out/ab_b1_$label.err
  rc=$?
  if grep -q "Memory access fault" gpurun_out/ab_b1_$label.err; then echo "GPU FAULT in $label"; exit 1; fi
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_b1_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_b1_$label.json').read().strip().splitlines()[-1])
print('$label', 'ms',round(d['ms_per_step'],2),{k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if '+' not in k}, d['checks'])"
}
for i in 1 2 3 4; do
run b10_$i KMU_COUNT_B1=10
run b11_$i KMU_X=1
done
