#!/bin/bash
# host leg: signature rows stored straight into the caller's pinned buffer (KMU_PIPE_ZC=1) against a download per chunk
cd $GRAFT_REPO_ROOT
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/ab_zc_$label.json 2> gpurun_out/ab_zc_$label.err
  rc=$?
  if grep -q "Memory access fault" gpurun_out/ab_zc_$label.err; then echo "GPU FAULT in $label"; exit 1; fi
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_zc_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_zc_$label.json').read().strip().splitlines()[-1])
print('$label', 'dev ms',round(d['ms_per_step'],2),'host ms', round(d['host_to_host']['ms_per_step'],2), d['checks'].get('host_leg_equals_device_leg'))"
}




run base KMU_X=1
run prio KMU_PIPE_D2H_PRIO=1
run late KMU_PIPE_DL=late
run base_b KMU_X=1
run prio_b KMU_PIPE_D2H_PRIO=1
run late_b KMU_PIPE_DL=late
