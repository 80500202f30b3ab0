#!/bin/bash
# needs a diagnostics build: KMU_BUILD_DEFS=-DKMU_DIAG=1 python kmerutils_amd/build.py --force
# diagnostics: scatter pass time against the partition fan-out (incomplete partitions, the table is not built)
cd $GRAFT_REPO_ROOT
SPLITS=${SPLITS:-11,10 10,10 9,9 8,8 7,7 6,6}
for sp in $SPLITS; do
  KMU_DBG_SPLIT=$sp timeout -k 10 120 python bench.py --workload ont_k31_count --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']; print('split','$sp', ' '.join('%s=%.2f'%(n[7:],k[n]['avg_ms']) for n in sorted(k)))"
done
