#!/bin/bash
# round 4: k_sketch_super with the signature mode / index draw as template constants (product) against looked up per step
# (libkmu_t0.so: scripts/build_variant.sh t0 "-DKMU_SUPER_TMODE=0" kmu_sketch_super)
cd $GRAFT_REPO_ROOT
for wl in c5_aa c1_super; do
  AB_LIBS="${SUPER_LIBS:-h t0 h t0}" AB_WORKLOAD=$wl bash scripts/r04_swar.sh | sed "s/^/$wl /"
done
