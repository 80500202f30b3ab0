#!/bin/bash
# diagnostics: the count-only bench on the product library (h) and variants kmerutils_amd/libkmu_<x>.so, same box, back to back
cd $GRAFT_REPO_ROOT
for x in ${AB_LIBS:-h}; do
  if [ "$x" != "h" ]; then export KMU_LIB=$PWD/kmerutils_amd/libkmu_$x.so; else unset KMU_LIB; fi
  timeout -k 10 200 python bench.py --workload ${AB_WORKLOAD:-ont_k31_count} --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg ${AB_ARGS} 2> gpurun_out/ab_count_$x.err | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('lib $x', round(d['ms_per_step'],2), {k: round(v['avg_ms'],2) for k,v in d['kernels'].items()}, d['checks'])" || { tail -5 gpurun_out/ab_count_$x.err; exit 1; }
  grep "diag seg" gpurun_out/ab_count_$x.err | tail -2
done
