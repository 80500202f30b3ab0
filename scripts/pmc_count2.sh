#!/bin/bash
# diagnostics: SQ + memory-path counters of the count pipeline kernels on the count-only bench workload
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
i=0
for set in ${PMC_SETS:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"}; do
  set=${set//,/ }
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_ct_$i -- python3 $R/bench.py --workload ont_k31_count --steps 1 --warmup 0 --no-cpu-baseline --no-host-leg --no-parity > $R/gpurun_out/pmc_ct_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ["GRAFT_REPO_ROOT"]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(R+"/gpurun_out/pmc_ct_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ","").replace("kmu::","").strip()
        if n.startswith("k_part") or n.startswith("k_arr"):
            acc[n][r["Counter_Name"]]+=float(r["Counter_Value"])
    os.remove(f)
for n in acc:
    print(n, " ".join("%s=%.4g"%(k.replace("SQ_",""),v) for k,v in sorted(acc[n].items())))
PY
