#!/bin/bash
# round 4: the closures selected once per launch (product library) against every key through apply_fhash (libkmu_f0.so:
# scripts/build_variant.sh f0 "-DKMU_UQ_FASTKEY=0" kmu_sketch), three workloads, same box, alternating
cd $GRAFT_REPO_ROOT
for wl in ont_k31_sketch c3_k8 short_k21_sketch; do
  AB_LIBS="h f0 h f0" AB_WORKLOAD=$wl bash scripts/r04_swar.sh | sed "s/^/$wl /"
done
