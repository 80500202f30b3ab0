#!/bin/bash
# phase shares of the two scatter levels (diagnostic build with cycle counters in thread 0 of every workgroup)
cd $GRAFT_REPO_ROOT
KMU_DIAG_SEG=1 KMU_LIB=$GRAFT_REPO_ROOT/kmerutils_amd/libkmu_d.so timeout -k 10 300 python bench.py --workload ont_k31_count --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-host-leg > gpurun_out/diagseg.json 2> gpurun_out/diagseg.err
rc=$?
grep -q "Memory access fault" gpurun_out/diagseg.err && { echo GPU FAULT; exit 1; }
[ $rc -eq 0 ] || { tail -5 gpurun_out/diagseg.err; exit 1; }
grep "diag seg" gpurun_out/diagseg.err | tail -4
python3 -c "
import json;d=json.loads(open('gpurun_out/diagseg.json').read().strip().splitlines()[-1])
print({k:round(v['avg_ms'],2) for k,v in d['kernels'].items()})"
