#!/bin/bash
# The N-rank code path on one rank (communicator to self): census, route, record scatter, all-to-all, build from what arrives.
# usage: scripts/r04_ncomm.sh [tag]   -> gpurun_out/ncomm_<tag>_*.json
tag=${1:-r04}
mkdir -p gpurun_out
for owner in minimizer hash; do
  for wl in ont_k31 c4_count; do
    KMU_BENCH_FORCE_COMM=1 KMU_COUNT_OWNER=$owner KMU_COUNT_ROUTE=occurrences python bench.py --workload $wl --steps 3 --warmup 1 --no-host-leg --no-cpu-baseline \
      > gpurun_out/ncomm_${tag}_${wl}_${owner}.json 2> gpurun_out/ncomm_${tag}_${wl}_${owner}.err || exit 1
    python - <<PY
import json
d = json.loads(open("gpurun_out/ncomm_${tag}_${wl}_${owner}.json").read().strip().splitlines()[-1])
print("${wl} ${owner}: %.1f ms/step" % d["ms_per_step"], {k: round(v["avg_ms"], 2) for k, v in d["kernels"].items() if "+" not in k}, d["comm"], d["checks"])
PY
  done
done
