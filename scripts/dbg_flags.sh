#!/bin/bash
# diagnostics: rebuild libkmu with extra compiler flags (KMU_BUILD_DEFS) and time the default bench kernels
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$TAG', round(d['ms_per_step'],2), {k: round(v['avg_ms'],2) for k,v in d['kernels'].items()})"
