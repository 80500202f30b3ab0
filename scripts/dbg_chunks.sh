#!/bin/bash
# needs a diagnostics build: KMU_BUILD_DEFS=-DKMU_DIAG=1 python kmerutils_amd/build.py --force
# diagnostics: level-2 scatter time against the number of units per level-1 partition
cd $GRAFT_REPO_ROOT
for c in ${CHUNKS:-1 2 4 8 16 32 64}; do
  KMU_DBG_CHUNKS2=$c timeout -k 10 120 python bench.py --workload ont_k31_count --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']; print('chunks2','$c', ' '.join('%s=%.2f'%(n[7:],k[n]['avg_ms']) for n in sorted(k)))"
done
