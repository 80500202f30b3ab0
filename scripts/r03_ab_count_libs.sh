#!/bin/bash
# same-box A/B of the count unit between library builds: h = the product library, others = kmerutils_amd/libkmu_<x>.so
cd $GRAFT_REPO_ROOT
for x in ${AB_LIBS:-h nt h nt h nt}; do
  if [ "$x" != "h" ]; then export KMU_LIB=$PWD/kmerutils_amd/libkmu_$x.so; else unset KMU_LIB; fi
  timeout -k 10 200 python bench.py --workload ont_k31_count --steps 4 --warmup 1 --no-cpu-baseline --no-host-leg > gpurun_out/ab_cl_$x.json 2> gpurun_out/ab_cl_$x.err; rc=$?
  grep -q "Memory access fault" gpurun_out/ab_cl_$x.err && { echo "GPU FAULT lib $x"; exit 1; }
  [ $rc -eq 0 ] || { echo "lib $x failed"; tail -3 gpurun_out/ab_cl_$x.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_cl_$x.json').read().strip().splitlines()[-1])
print('lib $x ms',round(d['ms_per_step'],2),{k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if '+' not in k},d['checks'])"
done
