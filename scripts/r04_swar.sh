#!/bin/bash
# round 4: the SWAR base packing and the rolled reverse complement of a wave step against the forms they replace
# (libkmu_a.so = both off, libkmu_b.so = packing only, built by scripts/build_variant.sh), same box, alternating
cd $GRAFT_REPO_ROOT
for x in ${AB_LIBS:-h a b h a b}; do
  if [ "$x" != "h" ]; then export KMU_LIB=$PWD/kmerutils_amd/libkmu_$x.so; else unset KMU_LIB; fi
  timeout -k 10 200 python bench.py --workload ${AB_WORKLOAD:-ont_k31} --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('lib $x', round(d['ms_per_step'],2), {k: round(v['avg_ms'],2) for k,v in d['kernels'].items()})" || exit 1
done
