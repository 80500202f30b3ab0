#!/bin/bash
# diagnostics: the sketch-only bench on library variants kmerutils_amd/libkmu_<x>.so (built by hand), same box, back to back
cd $GRAFT_REPO_ROOT
for x in ${AB_LIBS:-h a b h a b}; do
  if [ "$x" != "h" ]; then export KMU_LIB=$PWD/kmerutils_amd/libkmu_$x.so; else unset KMU_LIB; fi  # h = the product library
  timeout -k 10 120 python bench.py --workload ${AB_WORKLOAD:-ont_k31_sketch} --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('lib $x', {k: round(v['avg_ms'],2) for k,v in d['kernels'].items()})" || exit 1
done
