#!/usr/bin/env python3
"""Turn rocprofv3 output directories (gpurun_out/<name>_{trace,fetch,write}) into the small summaries committed
under profiles/: kernel stats (our kernels + the largest others) and per-launch HBM traffic from the PMC passes.

usage: scripts/summarize_prof.py <tag> <trace_dir> [<fetch_dir> <write_dir> [<insts_dir>]]
With --bench <bench line of the traced run> the PMC summary is also written to profiles/pmc_latest.json (what bench.py reads).
"""
import collections
import csv
import glob
import json
import os
import sys


# the kernels whose library timer (what bench.py groups into units) carries another name than the function rocprofv3 sees
TIMER_OF = {"k_arr_scatter_seg": "k_arr_scatter", "k_arr_scatter_exact": "k_arr_scatter", "k_part_scatter1_exact": "k_part_scatter1"}


def timer_name(kernel_name):
    """"void kmu::k_sketch_pmh3a<false, false>(kmu::SketchArgs)" -> "k_sketch_pmh3a" """
    k = kernel_name.split("(")[0].split("<")[0].replace("void ", "").replace("kmu::", "").strip()
    return TIMER_OF.get(k, k)


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    rows = list(csv.DictReader(open(glob.glob(os.path.join(trace, "*", "*_kernel_stats.csv"))[0])))
    with open(os.path.join(out_dir, "%s_kernel_stats.csv" % tag), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (kmu:: kernels + the 5 largest others)\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        others = 0
        for r in rows:
            ours = "kmu::" in r["Name"].split("(")[0]
            if ours or others < 5:
                others += 0 if ours else 1
                f.write('"%s",%s,%s,%s,%s,%s,%s\n' % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                     r["Percentage"], r["MinNs"], r["MaxNs"]))
    pmc = {}
    if len(sys.argv) >= 5:
        for name, d in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            acc = collections.defaultdict(float)
            cnt = collections.Counter()
            for r in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0])):
                if r["Counter_Name"] == name and "kmu::" in r["Kernel_Name"].split("(")[0]:
                    # "void kmu::k_sketch_pmh3a<false, false>(kmu::SketchArgs)" -> "k_sketch_pmh3a"
                    k = timer_name(r["Kernel_Name"])
                    acc[k] += float(r["Counter_Value"])
                    cnt[k] += 1
            for k in acc:
                pmc.setdefault(k, {})[name + "_KB_per_launch"] = acc[k] / cnt[k]
        for k, v in pmc.items():
            f_kb, w_kb = v.get("FETCH_SIZE_KB_per_launch", 0), v.get("WRITE_SIZE_KB_per_launch", 0)
            # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming
            # read -> doubled for the streaming kernels; the random 4/8-byte atomic pattern of the count-insert kernel
            # is uncalibrated and left as reported.
            streaming = not k.startswith("k_count_add")
            v["fetch_correction"] = 2.0 if streaming else 1.0
            v["hbm_bytes_per_launch"] = (f_kb * v["fetch_correction"] + w_kb) * 1024.0
        if len(sys.argv) >= 6 and not sys.argv[5].startswith("--"):
            # wave-instructions issued per launch (summed over the XCDs / shader engines the counter is sampled on)
            acc = collections.defaultdict(lambda: collections.defaultdict(float))
            cnt = collections.defaultdict(collections.Counter)
            for r in csv.DictReader(open(glob.glob(os.path.join(sys.argv[5], "*", "*_counter_collection.csv"))[0])):
                if "kmu::" in r["Kernel_Name"].split("(")[0]:
                    k = timer_name(r["Kernel_Name"])
                    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[k][r["Counter_Name"]] += 1
            for k in acc:
                for name in acc[k]:
                    pmc.setdefault(k, {})[name + "_per_launch"] = acc[k][name] / cnt[k][name]
        json.dump(pmc, open(os.path.join(out_dir, "%s_pmc.json" % tag), "w"), indent=1, sort_keys=True)
        if "--bench" in sys.argv:  # the bench line of the traced run names the workload the counters belong to
            b = json.loads(open(sys.argv[sys.argv.index("--bench") + 1]).readline())
            if "--no-latest" not in sys.argv:  # (pmc_latest.json is the headline workload's: what bench.py reads)
                json.dump({"round": tag, "source": "scripts/pmc_bench.sh", "workload": b["config"]["workload_name"],
                           "bases_per_gpu": b["config"]["bases_per_gpu"], "kernels": pmc},
                          open(os.path.join(out_dir, "pmc_latest.json"), "w"), indent=1, sort_keys=True)
            # a roofline object from THIS run's counters: the units bench.py priced, with the traffic and instruction counts
            # of the passes above (the bench line itself only knows the committed headline profile)
            units = {}
            for name, e in (b.get("kernels") or {}).items():
                if "alg_bytes" not in e:
                    continue
                parts = name.split("+")
                mult = (e.get("per_step") or {})
                alias = {"k_nthash": "k_nthash_flat"}  # (the library's timer name -> the kernel rocprofv3 saw)
                tr = [pmc.get(alias.get(k, k), {}).get("hbm_bytes_per_launch") for k in parts]
                va = [pmc.get(alias.get(k, k), {}).get("SQ_INSTS_VALU_per_launch") for k in parts]
                m = [mult.get(k, 1.0) for k in parts]
                units[name] = {"bound": "hbm", "achieved": e["GBps"], "peak": 8000.0, "unit": "GB/s", "frac": e["GBps"] / 8000.0,
                               "avg_launch_ms": e["avg_ms"], "alg_bytes_per_launch": e["alg_bytes"],
                               "traffic": sum(t * x for t, x in zip(tr, m)) if all(t is not None for t in tr) else None,
                               "valu_wave_insts": sum(v * x for v, x in zip(va, m)) if all(v is not None for v in va) else None}
                if units[name]["valu_wave_insts"]:
                    units[name]["valu_issue_frac"] = units[name]["valu_wave_insts"] / (e["avg_ms"] * 1e-3) / (1.03e9 * 4 * 256)
            b["roofline_from_this_profile"] = units
            json.dump(b, open(os.path.join(out_dir, "%s_bench.json" % tag), "w"))
    print(open(os.path.join(out_dir, "%s_kernel_stats.csv" % tag)).read())
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main()
