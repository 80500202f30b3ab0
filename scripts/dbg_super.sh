#!/bin/bash
# diagnostics: k_sketch_super time against workgroup size / staging chunk
cd $GRAFT_REPO_ROOT
for w in ${WORKLOADS:-c5_aa c1_super}; do
for cfg in ${CFGS:-256,2048 256,1024 128,1024 128,512 64,512 64,256}; do
  t=${cfg%,*}; c=${cfg#*,}
  KMU_SUPER_THREADS=$t KMU_SUPER_CHUNK=$c timeout -k 10 120 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w threads $t chunk $c', round(d['kernels']['k_sketch_super']['avg_ms'],3))"
done; done
