#!/bin/bash
# diagnostics: the sketch-only bench under different environment settings, same box, back to back
# usage: AB_ENVS="A=1;B=2 C=3" scripts/ab_env.sh   (settings separated by spaces, variables inside one by ';')
cd $GRAFT_REPO_ROOT
for e in "" ${AB_ENVS}; do
  env $(echo $e | tr ';' ' ') timeout -k 10 120 python bench.py --workload ${AB_WORKLOAD:-ont_k31_sketch} --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('env [$e]', d['ms_per_step'], {k: round(v['avg_ms'],2) for k,v in d['kernels'].items()})" || exit 1
done
