#!/bin/bash
# k_part_scatter1's two speeds: consecutive processes of the same command, each with the kernel's mean time,
# the addresses the driver gave the partition buffers, and (second half) rocprofv3 counter passes whose per-dispatch rows carry
# the dispatch's duration next to the counters.  usage: scripts/dbg_scatter1.sh <tag> [n_plain] [extra bench args]
tag=${1:-r03}; n=${2:-6}; shift 2
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sc1_$tag; mkdir -p $O; cd /tmp
BENCH="python3 $R/bench.py --workload ont_k31_count --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-host-leg"
(rocprofv3 -L 2>/dev/null || rocprofv3-avail list 2>/dev/null) > $O/counters_avail.txt 2>&1
grep -o "\(TCP\|TCC\|UTCL\|TCA\|GRBM\|SQ\)[A-Z0-9_a-z\[\]]*" $O/counters_avail.txt | sort -u > $O/counter_names.txt; wc -l $O/counter_names.txt
for i in $(seq 1 $n); do
  KMU_DIAG_BUFS=1 timeout -k 10 120 $BENCH "$@" > $O/plain_$i.json 2> $O/plain_$i.err || { echo "run $i failed"; tail -3 $O/plain_$i.err; exit 1; }
  python3 - $O/plain_$i.json $O/plain_$i.err <<'PY'
import json, sys, re
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
addr = {m.group(1): m.group(2) for m in re.finditer(r"kmu dev_buf (\S+)\s+(0x[0-9a-f]+)", open(sys.argv[2]).read())}
print("scatter1 %.2f  arr_scatter %.2f  build %.2f | partA %s partB %s" % (k["k_part_scatter1"]["avg_ms"], k["k_arr_scatter"]["avg_ms"],
      (k.get("k_part_build_q") or k["k_part_build"])["avg_ms"], addr.get("cnt.partA"), addr.get("cnt.partB")))
PY
done
# counter passes (one group per process; the same command): durations come from the kernel trace of the same run
pass() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$name -- $BENCH > $O/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $O/pmc_$name.log; return 0; }
  python3 - $O/pmc_$name <<'PY'
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
    if k not in ("k_part_scatter1", "k_arr_scatter", "k_part_build", "k_part_build_q"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"])); n[k] += 1; ms[k] += dur.get(r["Dispatch_Id"], 0.0)
for k in acc:
    print(k, "%.2f ms/launch" % (ms[k] / max(n[k], 1)), {c: "%.4g" % (v / n[k]) for c, v in acc[k].items()})
PY
  find $O/pmc_$name -name "*.csv" -size +2M -delete
}
for rep in 1 2; do
  pass utcl_$rep TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
  pass tcc_$rep TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum
  pass tcp_$rep TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
done
