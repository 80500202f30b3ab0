#!/bin/bash
# the host-to-host leg of the default bench under settings of the packed upload (KMU_PIPE_*): same box, back to back
# usage: scripts/r04_hostleg.sh "ENV1=a ENV2=b" "ENV1=c" ...
cd $GRAFT_REPO_ROOT
for setting in "$@"; do
  env $setting timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs --no-parity --steps 3 2>gpurun_out/hl.err | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('$setting: device %.1f ms, host to host %.1f ms = %.2f Gbases/s' % (d['ms_per_step'], d['host_to_host']['ms_per_step'], d['host_to_host']['value']), d['checks'].get('host_leg_equals_device_leg'))" || tail -3 gpurun_out/hl.err
done
