#!/bin/bash
# level 1 with a one-multiply stand-in for the table hash (variant abl1: wrong tables; how much of the level is the hash arithmetic)
cd $GRAFT_REPO_ROOT
for v in abl1 base abl1 base; do
  if [ $v = base ]; then L=""; else L=$GRAFT_REPO_ROOT/kmerutils_amd/libkmu_$v.so; fi
  KMU_LIB=$L timeout -k 10 200 python bench.py --workload ont_k31_count --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-host-leg > gpurun_out/abl1_$v.json 2> gpurun_out/abl1_$v.err
  python3 -c "
import json;d=json.loads(open('gpurun_out/abl1_$v.json').read().strip().splitlines()[-1])
print('$v', {k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if '+' not in k})" || tail -3 gpurun_out/abl1_$v.err
done
