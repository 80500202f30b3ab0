#!/bin/bash
# diagnostics: SQ counters of k_sketch_pmh3a on the sketch-only bench workload
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_sk_$tag -- python3 $R/bench.py --workload ont_k31_sketch --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_sk_$tag.log 2>&1
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in glob.glob(R+"/gpurun_out/pmc_sk_*/*/*_counter_collection.csv"):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_sketch_pmh3a" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in sorted(acc.items()): print(k, "%.4g"%v)
PY
