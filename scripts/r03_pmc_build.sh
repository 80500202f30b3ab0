#!/bin/bash
# counters of k_part_build in both slot formats (same box)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcb; mkdir -p $O; cd /tmp
BENCH="python3 $R/bench.py --workload ont_k31_count --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-host-leg"
pass() { # fmt name counters...
  fmt=$1; name=$2; shift 2
  KMU_COUNT_FMT=$fmt timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/${fmt}_$name -- $BENCH > $O/${fmt}_$name.log 2>&1 || { echo "pmc $fmt $name failed"; tail -3 $O/${fmt}_$name.log; return 0; }
  python3 - $O/${fmt}_$name $fmt <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + "/*/*_counter_collection.csv"); kt = glob.glob(d + "/*/*_kernel_trace.csv")
if not cc: print("no counters"); sys.exit(0)
dur = {}
for r in csv.DictReader(open(kt[0])) if kt else []:
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); ms = collections.defaultdict(float); seen = set()
for r in csv.DictReader(open(cc[0])):
    k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("kmu::", "").strip()
    if k not in ("k_part_build", "k_part_build_q"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"])); n[k] += 1; ms[k] += dur.get(r["Dispatch_Id"], 0.0)
for k in acc:
    print(sys.argv[2], k, "%.2f ms/launch" % (ms[k] / max(n[k], 1)), {c: "%.4g" % (v / n[k]) for c, v in acc[k].items()})
PY
  rm -rf $O/${fmt}_$name
}
for fmt in wide quot; do
  pass $fmt a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
  pass $fmt b SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE
  pass $fmt c SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  pass $fmt d SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM
done
