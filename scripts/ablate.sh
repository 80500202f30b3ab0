#!/bin/bash
# diagnostics: phase ablation of k_sketch_pmh3a on the bench read set (not part of the product)
cd $GRAFT_REPO_ROOT
# needs a diagnostics build: KMU_BUILD_DEFS=-DKMU_DIAG=1 python kmerutils_amd/build.py --force
for ab in ${ABLATE_LIST:-0 1 3 7}; do
  KMU_PMH_ABLATE=$ab timeout -k 10 120 python bench.py --workload ont_k31_sketch --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ablate',$ab, d['kernels']['k_sketch_pmh3a']['avg_ms'])"
done
