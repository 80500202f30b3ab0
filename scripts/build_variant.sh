#!/bin/bash
# diagnostics: build a library variant kmerutils_amd/libkmu_<tag>.so (select it with KMU_LIB=...): the named sources are
# recompiled with extra flags, the other objects come from the product build (kmerutils_amd/build/).
# usage: scripts/build_variant.sh <tag> "<extra hipcc flags>" <source> [<source> ...]      e.g.  d "-DKMU_DIAG=1" kmu_count
set -e
cd "$(dirname "$0")/.."
tag=$1; flags=$2; shift 2
python kmerutils_amd/build.py > /dev/null
mkdir -p kmerutils_amd/build_$tag
objs=""
for s in kmu_api kmu_sketch kmu_sketch_kernels kmu_sketch_super kmu_sketch_dens kmu_count kmu_count_part kmu_count_dist kmu_smer kmu_hostpack kmu_compare kmu_ingest kmu_kmergen kmu_comm; do
  if [[ " $* " == *" $s "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $flags \
      -c kmerutils_amd/csrc/$s.hip -o kmerutils_amd/build_$tag/$s.o &
    objs="$objs kmerutils_amd/build_$tag/$s.o"
  else
    objs="$objs kmerutils_amd/build/$s.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kmerutils_amd/libkmu_$tag.so $objs -ldl -lpthread
echo kmerutils_amd/libkmu_$tag.so
