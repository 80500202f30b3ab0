#!/bin/bash
# end-of-round measurement on one GPU box: the full -m gpu suite, the rocprofv3 / PMC passes of the default bench (profiles/<tag>_*),
# the default bench line with cpu_baseline, the other workloads.  usage: scripts/final_round.sh <tag>
# (a gpurun call is at most 20 minutes: `scripts/final_round.sh <tag> 1` = the suite + the headline's profile passes + the default
#  line, `... <tag> 2` = the other workloads, the routes, the N-rank path on one rank, the host leg)
tag=$1; stage=${2:-12}
cd $GRAFT_REPO_ROOT
if [[ $stage == *1* ]]; then
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t_final_$tag.log 2>&1; echo rc=$? >> gpurun_out/t_final_$tag.log; tail -3 gpurun_out/t_final_$tag.log
grep -q "^rc=0" gpurun_out/t_final_$tag.log || exit 1
bash scripts/pmc_bench.sh $tag > gpurun_out/pmc_bench_$tag.log 2>&1 || { echo "pmc_bench failed"; tail -5 gpurun_out/pmc_bench_$tag.log; exit 1; }
timeout -k 10 300 python bench.py > gpurun_out/bench_${tag}_default.json 2> gpurun_out/bench_${tag}_default.err || { echo "bench failed"; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/bench_${tag}_default.json').read().strip().splitlines()[-1])
print('default line: value', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'host to host', round(d['host_to_host']['ms_per_step'],2), round(d['metric_value_8d'],2), 'frac', round(d['roofline']['frac'],4), d['roofline']['kernel'])
print({k: round(v['avg_ms'],2) for k,v in d['kernels'].items()}); print(d['checks']); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for n,u in d['roofline']['units'].items(): print(n, 'frac %.4f' % u['frac'], 'traffic', u['traffic'], 'alu', (u.get('alu') or {}).get('frac'))"
fi
[[ $stage == *2* ]] || exit 0
WORKLOADS="c3_k8 c2_count c2_nthash_count c4_count c1_super c5_aa short_k21_sketch" bash scripts/other_workloads.sh
python3 - <<PY
import json
runs = {}
for w in "c3_k8 c2_count c2_nthash_count c4_count c1_super c5_aa short_k21_sketch".split():
    try:
        d = json.loads(open("gpurun_out/wl_%s.json" % w).read().strip().splitlines()[-1])
    except Exception as e:
        print("no line for", w, e); continue
    runs[w] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
               "value_host_to_host": d.get("value_host_to_host"), "kernels": {k: v["avg_ms"] for k, v in d["kernels"].items()},
               "checks": d["checks"], "cpu_baseline": (d.get("cpu_baseline") or {}).get("value"),
               # the HBM roofline of the workload's dominant unit as the line computed it (traffic / instruction counts of the
               # same workload: profiles/<round>a_<workload>_bench.json, roofline_from_this_profile)
               "roofline": {k: (d.get("roofline") or {}).get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "alg_bytes_per_launch")}}
json.dump({"round": "$tag", "source": "scripts/final_round.sh", "runs": runs}, open("gpurun_out/${tag}_workloads.json", "w"), indent=1, sort_keys=True)
PY
# the routes of a distributed add through a communicator of one rank (the default owners: minimizer; occurrences = super-k-mer records)
ROUTE_WORKLOADS="c4_count ont_k31_count" bash scripts/routes.sh > gpurun_out/${tag}_routes.txt 2>&1; tail -6 gpurun_out/${tag}_routes.txt
# the N-rank code path of the default bench on one rank, minimizer owners against hash owners
bash scripts/r04_ncomm.sh $tag > gpurun_out/${tag}_ncomm.txt 2>&1; cut -c1-400 gpurun_out/${tag}_ncomm.txt
# the host-to-host leg: packed upload (default), equal chunks, plain upload
bash scripts/r04_hostleg.sh "KMU_PIPE_GROWTH=3" "KMU_PIPE_GROWTH=1" "KMU_PIPE_PACK=0" > gpurun_out/${tag}_hostleg.txt 2>&1; cat gpurun_out/${tag}_hostleg.txt
