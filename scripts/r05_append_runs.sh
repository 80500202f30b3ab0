#!/bin/bash
# the write pattern of the single-pass partition in isolation (scripts/micro/append_runs.hip): time and the fabric's WRITE_SIZE per launch
# by run length and alignment.  usage (GPU box): scripts/r05_append_runs.sh  -> gpurun_out/r05_append_runs.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_append; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/append_runs scripts/micro/append_runs.hip || exit 1
out=gpurun_out/r05_append_runs.txt; : > $out
run() { # R mode sets bins bytes
  local tag="R$1_m$2_s$3_b$4_$5"
  timeout -k 10 120 $O/append_runs $1 $2 $3 $4 $5 > $O/$tag.txt 2>&1 || { echo "$tag failed" >> $out; return; }
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$tag -- $O/append_runs $1 $2 $3 $4 $5 > $O/$tag.log 2>&1
  python3 - $O/$tag "$(cat $O/$tag.txt)" >> $out <<'PY'
import sys, glob, csv
vals = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "WRITE_SIZE" and "k_append" in r.get("Kernel_Name", ""):
            vals.append(float(r["Counter_Value"]))
w = (sum(vals) / len(vals) * 1024 / 1e9) if vals else float("nan")
print(sys.argv[2], "| WRITE_SIZE %.2f GB per launch" % w)
PY
}
run 8 0 16 2048 8
run 8 1 16 2048 8
run 8 2 16 2048 8
run 8 3 16 2048 8
run 8 4 16 2048 8
run 8 0 8 2048 8
run 8 0 32 2048 8
run 4 0 16 2048 8
run 12 0 16 2048 8
run 16 0 16 1024 8
run 16 1 16 1024 8
run 24 0 16 724 6
run 24 1 16 724 6
run 32 0 16 512 8
run 64 0 16 256 8
cat $out
