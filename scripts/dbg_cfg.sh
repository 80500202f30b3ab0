#!/bin/bash
# diagnostics: k_sketch_pmh3a time under different workgroup shapes (KMU_PMH_SLOTS / KMU_PMH_THREADS)
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-0,1024 3584,512 3072,512 2560,512}; do
  s=${cfg%,*}; t=${cfg#*,}
  KMU_PMH_SLOTS=$s KMU_PMH_THREADS=$t timeout -k 10 120 python bench.py --workload ont_k31_sketch --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('slots $s threads $t', d['kernels']['k_sketch_pmh3a']['avg_ms'])"
done
