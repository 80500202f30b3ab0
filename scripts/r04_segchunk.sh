#!/bin/bash
# round 4: the shared level-1 streams CHUNKED (KMU_COUNT_SEG_CHUNK = log2 of the chunk in items; 0 = contiguous streams), same box
cd $GRAFT_REPO_ROOT
for x in ${SEGCHUNK_SEQ:-0 10 12 8 0 10 12 8}; do
  KMU_COUNT_SEG_CHUNK=$x timeout -k 10 200 python bench.py --workload ${SEGCHUNK_WL:-ont_k31_count} --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('chunk $x', round(d['ms_per_step'],2), {k: round(v['avg_ms'],2) for k,v in d['kernels'].items()}, d['checks'].get('count_conservation_ok'), d['checks'].get('parity_counts_ok'))" || exit 1
done
