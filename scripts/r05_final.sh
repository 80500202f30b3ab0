#!/bin/bash
# round 5, stage 2 of the end-of-round measurement: the COPY transport tests, the headline's PMC passes (profiles/<tag>_*), config 2's,
# the N-rank path on one rank by transport.  usage: scripts/r05_final.sh <tag>
tag=$1
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_comm.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/t_comm_$tag.log 2>&1; echo rc=$? >> gpurun_out/t_comm_$tag.log; tail -4 gpurun_out/t_comm_$tag.log
bash scripts/pmc_bench.sh $tag > gpurun_out/pmc_bench_$tag.log 2>&1 || { echo "pmc_bench failed"; tail -5 gpurun_out/pmc_bench_$tag.log; }
python3 - <<PY
import json
b = json.load(open("profiles/${tag}_bench.json"))
for n, u in b["roofline_from_this_profile"].items():
    print(n[:50], "%.2f ms" % u["avg_launch_ms"], "frac %.4f" % u["frac"], "traffic %.1f GB = %.2fx" % (u["traffic"] / 1e9, u["traffic"] / u["alg_bytes_per_launch"]) if u["traffic"] else "traffic n/a", "valu issue frac %.2f" % u.get("valu_issue_frac", 0))
p = json.load(open("profiles/${tag}_pmc.json"))
for k in ("k_part_scatter1", "k_arr_scatter", "k_part_build_q", "k_multiset_uq", "k_sketch_pmh3a", "k_pmh_points"):
    if k in p: print(k, "%.1f GB per launch" % (p[k]["hbm_bytes_per_launch"] / 1e9), "fetch %.1f write %.1f" % (p[k].get("FETCH_SIZE_KB_per_launch", 0) * 2 * 1024 / 1e9, p[k].get("WRITE_SIZE_KB_per_launch", 0) * 1024 / 1e9))
PY
bash scripts/pmc_workloads.sh $tag c2_nthash_count 2>&1 | tail -4
# the N-rank code path on one rank: by transport
for tr in copy rccl; do
  KMU_BENCH_FORCE_COMM=1 KMU_BENCH_TRANSPORT=$tr timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-host-leg > gpurun_out/${tag}_ncomm_$tr.json 2> gpurun_out/${tag}_ncomm_$tr.err
  python3 - $tr <<PY
import json,sys
tr=sys.argv[1]
d=json.loads(open("gpurun_out/${tag}_ncomm_%s.json" % tr).read().strip().splitlines()[-1])
c=d["comm"] or {}
print("N-rank path on one rank, transport %s (got %s): step %.2f ms, exchange_ms %.2f, checks %s" % (tr, c.get("transport"), d["ms_per_step"], (c.get("exchange_ms") or 0) / max(1, c.get("exchanges") or 1), {k: v for k, v in d["checks"].items() if k.startswith("parity") or k.startswith("count_")}))
print({k: round(v["avg_ms"], 2) for k, v in d["kernels"].items() if "+" not in k})
PY
done
