#!/bin/bash
# diagnostics: the sketch-only bench on library variants kmerutils_amd/libkmu_<x>.so (built by hand), same box, back to back
cd $GRAFT_REPO_ROOT
cp kmerutils_amd/libkmu.so /tmp/libkmu_orig.so
for x in ${AB_LIBS:-h a b h a b}; do
  cp kmerutils_amd/libkmu_$x.so kmerutils_amd/libkmu.so
  timeout -k 10 120 python bench.py --workload ${AB_WORKLOAD:-ont_k31_sketch} --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('lib $x', {k: round(v['avg_ms'],2) for k,v in d['kernels'].items()})" || exit 1
done
cp /tmp/libkmu_orig.so kmerutils_amd/libkmu.so
