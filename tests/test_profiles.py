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


def test_bench_starts_its_ranks_as_a_child_process(monkeypatch, capsys):
    """`python bench.py --gpus N` without a rank environment: N ranks through a child `torch.distributed.run` on 127.0.0.1,
    the arguments passed on, nothing of this process touching the GPU first; the child's JSON line is relayed"""
    import bench
    seen = []

    def fake_runner(cmd, env):
        seen.append((cmd, env))
        return 0, 'noise\n{"metric": "m", "value": 1.0, "n_gpus": 8}\n'

    monkeypatch.setattr(bench, "_run_child", fake_runner)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import pytest
    monkeypatch.setattr(bench.launch_ranks, "__defaults__", (None, fake_runner, None))
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 0 and len(seen) == 1
    cmd, env = seen[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert cmd[-5].endswith("bench.py") and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and "fallback_from" not in line
    assert "torch" not in {m for m in ("torch.cuda",) if getattr(sys.modules.get(m), "is_initialized", lambda: False)()}


def test_bench_falls_back_to_a_fresh_child_when_the_first_contact_fails(capsys):
    """the first N-GPU run must leave a record: a child that dies (or prints no JSON line) is followed by a FRESH child with the
    next transport of the chain copy -> rccl -> torch, then by one without a collective; the line names what failed
    (VERDICT r03 next #2b, r04 next #4)"""
    import bench
    args = bench.argparse.Namespace(gpus=8, workload="ont_k31")
    calls = []

    def runner(cmd, env):
        calls.append((cmd, env.get("KMU_BENCH_TRANSPORT")))
        if len(calls) == 1:
            return 134, "the copy transport blew up\n"  # killed: no JSON
        if len(calls) == 2:
            return 124, ""  # hung, ended by the attempt's timeout
        if len(calls) == 3:
            return 0, "partial output, no json\n"  # exits 0 but printed nothing usable
        return 0, '{"metric": "m", "value": 2.0, "n_gpus": 8, "config": {"workload_name": "ont_k31_sketch"}}\n'

    rc = bench.launch_ranks(args, argv=["--gpus", "8", "--workload", "ont_k31", "--steps", "2"], runner=runner, environ={"PATH": os.environ["PATH"]})
    assert rc == 0 and len(calls) == 4
    assert [t for _, t in calls] == ["copy", "rccl", "torch", None]
    assert calls[0][0][-6:] == ["--gpus", "8", "--workload", "ont_k31", "--steps", "2"]
    assert calls[3][0][-2:] == ["--workload", "ont_k31_sketch"] and calls[3][0].count("--workload") == 1
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["value"] == 2.0 and [f["exit_code"] for f in line["fallback_from"]] == [134, 124, 0]
    assert line["fallback_from"][2]["json_line"] is False and "no collective" in line["attempt"]
    # a substituted workload is no headline: the metric of the line says so
    assert line["workload_substituted"] == {"asked": "ont_k31", "ran": "ont_k31_sketch"} and "SUBSTITUTED" in line["metric"]
    # every attempt fails: a non-zero exit code, no line
    rc = bench.launch_ranks(args, argv=["--gpus", "8"], runner=lambda c, e: (9, ""), environ={"PATH": os.environ["PATH"]})
    assert rc == 9 and capsys.readouterr().out.strip() == ""
    # KMU_BENCH_TRANSPORT names where the chain starts; a workload without a collective has nothing to fall back from but the transport
    calls.clear()
    args2 = bench.argparse.Namespace(gpus=4, workload="c5_aa")
    assert bench.launch_ranks(args2, argv=["--gpus", "4", "--workload", "c5_aa"], runner=lambda c, e: (calls.append(e.get("KMU_BENCH_TRANSPORT")), (1, ""))[1],
                              environ={"PATH": os.environ["PATH"], "KMU_BENCH_TRANSPORT": "torch"}) == 1 and calls == ["torch"]
    calls.clear()
    assert bench.launch_ranks(args2, argv=["--gpus", "4", "--workload", "c5_aa"], runner=lambda c, e: (calls.append(e.get("KMU_BENCH_TRANSPORT")), (1, ""))[1],
                              environ={"PATH": os.environ["PATH"], "KMU_BENCH_TRANSPORT": "rccl"}) == 1 and calls == ["rccl", "torch"]


def test_a_hung_attempt_is_ended_with_its_process_group():
    """bench._run_child with a timeout: the child (and what it started) is gone afterwards, the attempt reads as failed (124)"""
    import bench
    import time
    t0 = time.time()
    rc, out = bench._run_child([sys.executable, "-c", "import subprocess, sys, time; print('started', flush=True); subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(60)']); time.sleep(60)"],
                               dict(os.environ), timeout=2.0)
    assert rc == 124 and "started" in out and time.time() - t0 < 30
    rc, out = bench._run_child([sys.executable, "-c", "print('{\"a\": 1}')"], dict(os.environ), timeout=30.0)
    assert rc == 0 and bench._last_json_line(out) == {"a": 1}

def test_the_default_run_times_every_baseline_configuration():
    """`configs` of the default one-GPU line: one child `python bench.py --workload <w>` per BASELINE
    configuration, each relayed as ms per step, throughput, roofline fraction of its dominant unit and oracle parity of its timed
    output; a child that fails leaves an `error` entry, not a missing key and not a dead headline."""
    import bench
    seen = []

    def runner(cmd, env):
        w = cmd[cmd.index("--workload") + 1]
        seen.append(cmd)
        if w == "c4_count":
            return 1, "boom\n"
        line = {"metric": "m", "value": 100.0, "ms_per_step": 7.5, "steps": 5, "warmup": 3,
                "config": {"workload": "%s: %s" % (w, "625000 protein sequences" if w == "c5_aa" else "1000000 x 150 bp reads")},
                "roofline": {"frac": 0.2, "kernel": "k_x", "avg_launch_ms": 7.0},
                "checks": {"parity_reads": 1000, "parity_rows_ok": True, "parity_counts_ok": w != "c2_nthash_count", "count_conservation_ok": True}}
        return 0, "noise\n" + json.dumps(line) + "\n"

    out = bench.run_configs(runner=runner, environ={"PATH": os.environ["PATH"]})
    assert set(out) == set(bench.CONFIG_WORKLOADS) == {"c1_super", "c2_nthash_count", "c3_k8", "c4_count", "c5_aa"}
    for cmd in seen:  # a plain one-GPU child each: no cpu baseline, no nested configs, 3 warm-up + 5 steps
        assert cmd[1].endswith("bench.py") and "--no-configs" in cmd and "--no-cpu-baseline" in cmd
        assert cmd[cmd.index("--steps") + 1] == "5" and cmd[cmd.index("--warmup") + 1] == "3" and cmd[cmd.index("--gpus") + 1] == "1"
    assert "error" in out["c4_count"] and "exit code 1" in out["c4_count"]["error"]
    for w in ("c1_super", "c3_k8", "c5_aa", "c2_nthash_count"):
        e = out[w]
        assert {"workload", "ms_per_step", "value", "unit", "roofline_frac", "parity_ok", "steps", "warmup"} <= set(e)
        assert e["ms_per_step"] == 7.5 and e["roofline_frac"] == 0.2 and e["steps"] == 5 and e["warmup"] == 3
    assert out["c5_aa"]["unit"] == "Gresidues/s" and out["c3_k8"]["unit"] == "Gbases/s"
    assert out["c1_super"]["parity_ok"] is True and out["c2_nthash_count"]["parity_ok"] is False  # every parity_* check must hold
