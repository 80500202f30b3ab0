#!/bin/bash
# same-box A/B of the count unit between environment settings: usage AB="label:ENV=V,ENV2=V2 label2:..." scripts/r03_ab_env.sh [workload]
cd $GRAFT_REPO_ROOT
wl=${1:-ont_k31_count}
for spec in $AB; do
  label=${spec%%:*}; envs=${spec#*:}; envs=${envs//,/ }
  env $envs timeout -k 10 200 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-host-leg > gpurun_out/ab_env_$label.json 2> gpurun_out/ab_env_$label.err; rc=$?
  grep -q "Memory access fault" gpurun_out/ab_env_$label.err && { echo "GPU FAULT in $label"; exit 1; }
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_env_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_env_$label.json').read().strip().splitlines()[-1])
print('$label', 'ms',round(d['ms_per_step'],2),{k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if '+' not in k}, {k:v for k,v in d['checks'].items() if k.endswith('_ok')})"
done
