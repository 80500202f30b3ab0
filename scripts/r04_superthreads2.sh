#!/bin/bash
# the new defaults of k_sketch_super (256 threads, 512 staged items) against round 1's (64, 256): configs 5 and 1, same box
cd $GRAFT_REPO_ROOT
for wl in c5_aa c1_super; do
for cfg in "0 0" "64 256" "0 0" "64 256"; do
  set -- $cfg
  if [ $1 != 0 ]; then export KMU_SUPER_THREADS=$1 KMU_SUPER_CHUNK=$2; else unset KMU_SUPER_THREADS KMU_SUPER_CHUNK; fi
  timeout -k 10 200 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$wl threads $1 chunk $2', round(d['ms_per_step'],3), d['checks'].get('parity_rows_ok'))" || exit 1
done; done
