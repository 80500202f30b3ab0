#!/bin/bash
# host-to-host leg A/B: taper on / off, chunk sizes
cd $GRAFT_REPO_ROOT
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/ab_host_$label.json 2> gpurun_out/ab_host_$label.err
  rc=$?
  if grep -q "Memory access fault" gpurun_out/ab_host_$label.err; then echo "GPU FAULT in $label"; exit 1; fi
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_host_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_host_$label.json').read().strip().splitlines()[-1])
print('$label', 'dev ms',round(d['ms_per_step'],2),'host ms', round(d['host_to_host']['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if '+' not in k}, d['checks'].get('host_leg_equals_device_leg'))"
}
run taper1 KMU_X=1
run taper0 KMU_PIPE_TAPER=0
run taper1_c256 KMU_PIPE_CHUNK_MB=256
run taper1_c1024 KMU_PIPE_CHUNK_MB=1024
run taper1b KMU_X=1
run taper0b KMU_PIPE_TAPER=0
