#!/bin/bash
# diagnostics (needs a KMU_DIAG build): VALU / SALU instruction counts of k_sketch_pmh3a under phase ablation
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for ab in ${ABLATE_LIST:-0 8 1 3 7 39}; do
  KMU_PMH_ABLATE=$ab timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc_ab_$ab -- python3 $R/bench.py --workload ont_k31_sketch --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_ab_$ab.log 2>&1 || echo "pass $ab failed"
  python3 - $ab <<'PY'
import csv,glob,os,collections,sys
R=os.environ["GRAFT_REPO_ROOT"]; ab=sys.argv[1]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_ab_%s/*/*_counter_collection.csv"%ab)):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_sketch_pmh3a" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    print("ablate", ab, " ".join("%s=%.4g"%(k.replace("SQ_INSTS_",""),v) for k,v in sorted(acc.items())))
    os.remove(f)
PY
done
