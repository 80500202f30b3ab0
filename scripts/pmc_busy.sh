#!/bin/bash
# diagnostics: how busy the vector / scalar ALUs are under k_sketch_pmh3a (sketch-only bench workload), + the issue cost table
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
if [ -z "$SKIP_RATES" ]; then hipcc --offload-arch=gfx950 -O3 $R/scripts/micro/valu_rates.hip -o /tmp/valu_rates && timeout -k 10 120 /tmp/valu_rates > $R/gpurun_out/valu_rates.txt 2>&1 || exit 1; fi
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "VALUBusy SALUBusy" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_busy_$i -- python3 $R/bench.py --workload ont_k31_sketch --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_busy_$i.log 2>&1
  rc=$?
  if [ $rc -ge 124 ]; then echo "pass $i killed ($rc)"; exit 1; fi
  [ $rc -ne 0 ] && echo "pass $i failed ($rc)"
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_busy_*/*/*_counter_collection.csv")):
    acc=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        for kn in ("k_sketch_pmh3a", "k_pmh_points"):
            if kn in r["Kernel_Name"]:
                acc[(kn, r["Counter_Name"])]+=float(r["Counter_Value"]); n[(kn, r["Counter_Name"])]+=1
    for k,v in sorted(acc.items()): print(k[0], k[1], "%.5g"%v, "rows", n[k])
    os.remove(f)
PY
[ -z "$SKIP_RATES" ] && cat $R/gpurun_out/valu_rates.txt
