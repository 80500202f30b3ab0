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


def test_kernel_units_group_the_step_into_a_sketch_and_a_count_unit():
    """bench.kernel_units (CPU, no GPU): the kernels of a step are grouped into the sketch unit and the count unit, each priced
    with SURVEY 8(d)'s algorithmic bytes over the summed launch time of its kernels in one step; the longer unit is the
    dominant one; single kernels of a unit carry design_bytes, never alg_bytes."""
    import bench
    steps = 5
    # (launches, total ms) over 5 steps, shaped like the headline workload's profile
    stats = {"k_multiset_uq": (10, 10 * 13.0), "k_sketch_pmh3a": (5, 5 * 6.3), "k_pmh_points": (5, 5 * 18.5),
             "k_part_scatter1": (5, 5 * 19.0), "k_arr_scatter": (5, 5 * 23.0), "k_part_build_q": (5, 5 * 23.0), "k_count_add_spill": (5, 0.05),
             "k_max_len": (5, 0.25)}
    bases, nk, n_reads, m = 4_379_626_696, 4_357_236_706, 746_333, 200
    kern, dom, per_step = bench.kernel_units(stats, steps, bases, nk, n_reads, m, 8, (1 << 33) * 8)
    sk, ct = "k_multiset_uq+k_sketch_pmh3a+k_pmh_points", "k_part_scatter1+k_arr_scatter+k_part_build_q+k_count_add_spill"
    assert dom == ct and set(per_step) == set(ct.split("+")) and per_step["k_arr_scatter"] == 1.0
    assert kern[sk]["alg_bytes"] == bases + n_reads * m * 8 and kern[ct]["alg_bytes"] == bases + 16 * nk
    assert abs(kern[sk]["avg_ms"] - (2 * 13.0 + 6.3 + 18.5)) < 1e-9 and kern[sk]["per_step"]["k_multiset_uq"] == 2.0
    assert abs(kern[ct]["avg_ms"] - (19.0 + 23.0 + 23.0 + 0.01)) < 1e-9
    assert abs(kern[ct]["GBps"] - kern[ct]["alg_bytes"] / (kern[ct]["avg_ms"] * 1e-3) / 1e9) < 1e-6
    for name in ("k_part_scatter1", "k_arr_scatter", "k_part_build_q", "k_pmh_points"):
        assert "alg_bytes" not in kern[name] and kern[name]["design_bytes"] > 0
    assert "design_bytes" not in kern["k_max_len"] and "alg_bytes" not in kern["k_max_len"]
    # sketch only: the sketch unit is the dominant one
    kern2, dom2, _ = bench.kernel_units({k: v for k, v in stats.items() if k in sk.split("+")}, steps, bases, nk, n_reads, m, 8, 0)
    assert dom2 == sk


def test_bench_starts_its_ranks_as_a_child_process(monkeypatch):
    """`python bench.py --gpus N` without a rank environment: N ranks through a child `torch.distributed.run` on 127.0.0.1,
    the arguments passed on, nothing of this process touching the GPU first"""
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import pytest
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert cmd[-5].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in {m for m in ("torch.cuda",) if getattr(sys.modules.get(m), "is_initialized", lambda: False)()}
