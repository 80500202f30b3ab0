#!/bin/bash
# round 4: k_sketch_super's workgroup size / staging chunk on config 5's shard (KMU_SUPER_THREADS, KMU_SUPER_CHUNK), same box
cd $GRAFT_REPO_ROOT
for cfg in "64 256" "128 256" "256 256" "128 512" "256 512" "256 1024" "64 256"; do
  set -- $cfg
  KMU_SUPER_THREADS=$1 KMU_SUPER_CHUNK=$2 timeout -k 10 200 python bench.py --workload c5_aa --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('threads $1 chunk $2', round(d['ms_per_step'],2), d['checks'])" || exit 1
done
