#!/bin/bash
# diagnostics: instruction / wait counters of the sketch kernels for a library variant (kmerutils_amd/libkmu_<x>.so)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for x in ${AB_LIBS:-h n}; do
  # the variant is selected by KMU_LIB (kmerutils_amd/lib.py); the product library is never overwritten
  if [ "$x" != "h" ]; then export KMU_LIB=$R/kmerutils_amd/libkmu_$x.so; else unset KMU_LIB; fi
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_lib_${x}_$i -- python3 $R/bench.py --workload ont_k31_sketch --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_lib_${x}_$i.log 2>&1 || exit 1
  done
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_lib_*/*/*_counter_collection.csv")):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_sketch_pmh3a" in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(f.split("/")[-3], {k:"%.4g"%v for k,v in sorted(acc.items())})
    os.remove(f)
PY
