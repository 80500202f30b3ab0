#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload ont_k31_count --steps 4 --warmup 1 --no-cpu-baseline --no-host-leg --no-parity > gpurun_out/ab_build_$label.json 2> gpurun_out/ab_build_$label.err || { echo "$label failed"; tail -3 gpurun_out/ab_build_$label.err; return; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_build_$label.json').read().strip().splitlines()[-1])
print('$label', 'ms',round(d['ms_per_step'],2),{k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if '+' not in k}, d['checks'])"
  grep "build clocks" gpurun_out/ab_build_$label.err | tail -1
}
run quot_lanes KMU_X=1
run quot_items KMU_BUILD_ABLATE=32
run quot_lanes_clk KMU_BUILD_ABLATE=16
run wide KMU_COUNT_FMT=wide
run quot_lanes2 KMU_X=1
