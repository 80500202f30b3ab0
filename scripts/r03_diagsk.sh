#!/bin/bash
# phase shares of the sketch kernels (diagnostic build, thread-0 clocks): the general list-emitting kernel on the reads beyond 20 480 k-mers
cd $GRAFT_REPO_ROOT
KMU_PMH_ABLATE=256 KMU_LIB=$GRAFT_REPO_ROOT/kmerutils_amd/libkmu_ds.so timeout -k 10 300 python bench.py --workload ont_k31_sketch --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-host-leg > gpurun_out/diagsk.json 2> gpurun_out/diagsk.err
rc=$?
grep -q "Memory access fault" gpurun_out/diagsk.err && { echo GPU FAULT; exit 1; }
[ $rc -eq 0 ] || { tail -5 gpurun_out/diagsk.err; exit 1; }
grep "kmu" gpurun_out/diagsk.err | tail -8
python3 -c "
import json;d=json.loads(open('gpurun_out/diagsk.json').read().strip().splitlines()[-1])
print({k:round(v['avg_ms'],2) for k,v in d['kernels'].items()})"
