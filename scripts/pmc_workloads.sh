#!/bin/bash
# rocprofv3 --kernel-trace --stats + the PMC passes (FETCH_SIZE, WRITE_SIZE, instruction counts; separate runs, the program
# directly behind `--`) for the BASELINE configurations other than the headline: profiles/<tag>_<workload>_{kernel_stats.csv,pmc.json,bench.json}.
# usage: scripts/pmc_workloads.sh <tag> [workloads...]
tag=$1; shift
W=${@:-"c3_k8 c5_aa c1_super c2_nthash_count c4_count"}
cd $GRAFT_REPO_ROOT
for w in $W; do
  PMC_NO_LATEST=1 bash scripts/pmc_bench.sh ${tag}_$w --workload $w > gpurun_out/pmc_${tag}_$w.log 2>&1 || { echo "pmc passes of $w failed"; tail -5 gpurun_out/pmc_${tag}_$w.log; exit 1; }
  python3 - <<PY
import json
b = json.load(open("profiles/${tag}_${w}_bench.json"))
for n, u in b["roofline_from_this_profile"].items():
    print("$w", n, "%.2f ms" % u["avg_launch_ms"], "frac %.4f" % u["frac"], "traffic/alg %.2f" % (u["traffic"] / u["alg_bytes_per_launch"]) if u["traffic"] else "traffic n/a",
          "valu issue frac %.2f" % u.get("valu_issue_frac", 0))
PY
done
