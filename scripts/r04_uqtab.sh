#!/bin/bash
# round 4: the collision groups of k_multiset_uq through a table (KMU_PMH_UQTAB=1) against the counting sort (0), same box, alternating
cd $GRAFT_REPO_ROOT
for x in ${UQTAB_SEQ:-0 1 0 1}; do
  KMU_PMH_UQTAB=$x timeout -k 10 200 python bench.py --workload ont_k31_sketch --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('uqtab $x', round(d['ms_per_step'],2), {k: round(v['avg_ms'],2) for k,v in d['kernels'].items()}, d['checks'].get('parity_rows_ok'))" || exit 1
done
