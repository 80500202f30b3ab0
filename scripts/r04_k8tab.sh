#!/bin/bash
# round 4: config 3 as written (k = 8) with the first points of k_pmh_points from a table of all 4^8 k-mers (KMU_PMH_K8TAB=1, opt-in) against
# drawn per key (KMU_PMH_K8TAB=0), same box, alternating
cd $GRAFT_REPO_ROOT
for x in ${K8TAB_SEQ:-1 0 1 0}; do
  KMU_PMH_K8TAB=$x timeout -k 10 200 python bench.py --workload c3_k8 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('k8tab $x', round(d['ms_per_step'],2), round(d['value'],1), {k: round(v['avg_ms'],2) for k,v in d['kernels'].items()}, d['checks'])" || exit 1
done
