#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_golden_signatures.py tests/test_cpp_mirror.py -q -x -k "super or aa or all_seqs or golden or sweep or mirror or partial" > gpurun_out/t_r03i.log 2>&1; rc=$?; echo rc=$rc >> gpurun_out/t_r03i.log; tail -4 gpurun_out/t_r03i.log
grep -q "Memory access fault" gpurun_out/t_r03i.log && { echo "GPU FAULT in the tests"; exit 1; }
[ $rc -eq 0 ] || exit 1
for wl in c5_aa c1_super; do
for x in h oldsup h oldsup; do
  if [ "$x" != "h" ]; then export KMU_LIB=$PWD/kmerutils_amd/libkmu_$x.so; else unset KMU_LIB; fi
  timeout -k 10 200 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg > gpurun_out/ab_sup_$x.json 2> gpurun_out/ab_sup_$x.err; rc=$?
  grep -q "Memory access fault" gpurun_out/ab_sup_$x.err && { echo "GPU FAULT lib $x"; exit 1; }
  [ $rc -eq 0 ] || { echo "lib $x failed"; tail -3 gpurun_out/ab_sup_$x.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_sup_$x.json').read().strip().splitlines()[-1])
print('$wl lib $x ms',round(d['ms_per_step'],3),{k:round(v['avg_ms'],3) for k,v in d['kernels'].items() if '+' not in k},d['checks'])"
done
done
