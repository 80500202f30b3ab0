#!/bin/bash
# diagnostics: SQ / instruction-cache counters of k_sketch_pmh3a on the sketch-only bench workload
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_sk_$i -- python3 $R/bench.py --workload ont_k31_sketch --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_sk_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_sk_*/*/*_counter_collection.csv")):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_sketch_pmh3a" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in sorted(acc.items()): print(k, "%.4g"%v)
    os.remove(f)
PY
