"""The committed PMC summary is what bench.py reads for `roofline.traffic` and `alu`: it has to describe the kernels the
bench's dominant unit is made of (CPU test: no GPU, no oracle)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_pmc_latest_matches_the_bench_line():
    import bench
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    assert d["workload"] == "ont_k31"
    pair = ("k_multiset_uq", "k_sketch_pmh3a", "k_pmh_points")
    for k in pair:
        assert d["kernels"][k]["hbm_bytes_per_launch"] > 0 and d["kernels"][k]["SQ_INSTS_VALU_per_launch"] > 0
    cfg = {"name": "ont_k31"}
    total = d["bases_per_gpu"]
    t = bench.pmc_traffic(cfg, total, "+".join(pair))
    assert t == sum(d["kernels"][k]["hbm_bytes_per_launch"] for k in pair)
    alu = bench.pmc_alu(cfg, total, "+".join(pair), 53.0)
    assert alu["valu_wave_insts"] == sum(d["kernels"][k]["SQ_INSTS_VALU_per_launch"] for k in pair)
    # a name launched twice a step (the two shapes of k_multiset_uq) counts twice: per-launch mean x launches per step
    two = {"k_multiset_uq": 2.0}
    assert bench.pmc_traffic(cfg, total, "+".join(pair), two) == t + d["kernels"]["k_multiset_uq"]["hbm_bytes_per_launch"]
    alu = bench.pmc_alu(cfg, total, "+".join(pair), 53.0, two)
    assert 0.3 < alu["frac"] < 1.0
    # another workload or size: no figure rather than a wrong one
    assert bench.pmc_traffic({"name": "c3_k8"}, total, pair[0]) is None
    assert bench.pmc_traffic(cfg, total // 2, pair[0]) is None


def test_kernel_stats_of_the_current_round_list_both_sketch_kernels():
    tag = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))["round"]
    txt = open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")).read()
    assert "k_multiset_uq<512" in txt and "k_multiset_uq<1024" in txt and "k_sketch_pmh3a<" in txt and "k_pmh_points" in txt
