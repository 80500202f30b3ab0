#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X: Gbases/s of the k-mer + sketch (+ count) hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, or plain
     `python bench.py --gpus N`, which starts the same N ranks as a child process)

A "step" is one pass of the hot path over the whole synthetic read set resident in HBM:
    default workload `ont_k31`: 746 333 ONT-shaped reads / 4.38 Gbases per GPU (BASELINE config 3's read set) at the
    metric's k = 31: ProbMinHash3a, 200 sketches per read (u64 signatures, fhash = int64_hash(min(kmer, revcomp)))
    + kmercount of every canonical 31-mer (table reset included), + for N > 1 the owner-partition exchange.
Weak scaling: every rank holds its own read shard of that size (same genome, different read seed); value is the
whole-job aggregate.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ont_k31", choices=list(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank holds a full-size shard; strong: the workload's reads are split over the ranks")
    ap.add_argument("--reads", type=int, default=0, help="override reads per GPU")
    ap.add_argument("--bases", type=float, default=0, help="override total bases per GPU")
    ap.add_argument("--genome", type=int, default=0)
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="reads timed on the host oracle (0 = auto)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the oracle baseline (0 = all the box has)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-to-host leg (pinned host buffers in and out)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed output")
    ap.add_argument("--sketch-size", type=int, default=0)
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the timed lines of the other BASELINE configurations (`configs` of the default one-GPU run)")
    return ap.parse_args()


WORKLOADS = ("ont_k31", "ont_k31_sketch", "ont_k31_count", "c3_k8", "c2_count", "c2_nthash_count", "c4_count", "c1_super",
             "c5_aa", "ont_k31_optdens", "short_k21_sketch")


def workload_cfg(args):
    from kmerutils_amd import _abi as A
    w = args.workload
    cfg = dict(name=w, n_reads=746_333, total_bases=4.38e9, fixed_len=None, errors=(0.04, 0.02, 0.02), k=31,
               kmer_type=A.KMER64BIT, m=200, algo=A.ALGO_PROB3A, sig=A.SIG_U64, hasher=A.HASHER_NOHASH,
               fhash=A.FHASH_CANON_INVHASH, sketch=True, count=True, nthash=False, seed=0xC3, genome=100_000_000)
    if w == "ont_k31_sketch":
        cfg.update(count=False)
    elif w == "ont_k31_count":
        cfg.update(sketch=False)
    elif w == "ont_k31_optdens":  # the same reads through OptDensHashSketch (one-permutation hashing, f64 bins)
        cfg.update(count=False, algo=A.ALGO_OPTDENS, sig=A.SIG_F64)
    elif w == "c3_k8":
        cfg.update(k=8, kmer_type=A.KMER32BIT, sig=A.SIG_U32, count=False)
    elif w in ("c2_count", "c2_nthash_count"):
        # config 2: 1 M x 150 bp from a 10 Mbp genome, k = 21; as written it is "ntHash + unique-filter count": the canonical
        # ntHash of every 21-mer (kmu_nthash, 8 bytes per position) and the count of every canonical 21-mer
        cfg.update(n_reads=1_000_000, total_bases=1.5e8, fixed_len=150, errors=(0.005, 0, 0), k=21, sketch=False,
                   nthash=w == "c2_nthash_count", seed=0xC2, genome=10_000_000)
    elif w == "short_k21_sketch":  # config 2's reads (1 M x 150 bp) through ProbMinHash3a: the short-read regime of the sketch kernels
        cfg.update(n_reads=1_000_000, total_bases=1.5e8, fixed_len=150, errors=(0.005, 0, 0), k=21, count=False, seed=0xC2,
                   genome=10_000_000)
    elif w == "c4_count":  # config 4's per-GPU shard: 6.25 M x 150 bp of the 50 M reads, 100 Mbp genome, k = 31, count only
        cfg.update(n_reads=6_250_000, total_bases=9.375e8, fixed_len=150, errors=(0.005, 0, 0), k=31, sketch=False, seed=0xC4,
                   strong_total_reads=50_000_000)
    elif w == "c1_super":
        cfg.update(n_reads=10_000, total_bases=1e7, fixed_len=1000, errors=(0, 0, 0), k=16, kmer_type=A.KMER16B32BIT,
                   m=64, algo=A.ALGO_SUPER, sig=A.SIG_F64, count=False, seed=0xC1, genome=20_000_000)
    elif w == "c5_aa":  # config 5, one GPU's share of the 5 M proteins
        cfg.update(n_reads=625_000, total_bases=2.1e8, k=12, kmer_type=A.KMERAA64BIT, m=128, algo=A.ALGO_SUPER,
                   sig=A.SIG_F64, fhash=A.FHASH_VALUE_MASKED, count=False, seed=0xC5, protein=True, strong_total_reads=5_000_000)
    if args.genome:
        cfg["genome"] = args.genome
    if args.reads:  # (with --scaling strong: the JOB's reads, in place of BASELINE's total)
        cfg["total_bases"] = cfg["total_bases"] * args.reads / cfg["n_reads"]
        cfg["n_reads"] = args.reads
        cfg.pop("strong_total_reads", None)
    if args.bases:
        cfg["total_bases"] = args.bases
    if args.sketch_size:
        cfg["m"] = args.sketch_size
    return cfg


def _run_child(cmd, env, timeout=None):
    """one attempt: the child's exit code and what it wrote to stdout (its stderr goes straight through).  A child that is still
    running after `timeout` seconds is ended with its whole process group (the ranks are grandchildren) and counts as failed:
    exit code 124, like timeout(1)."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout)
        return p.returncode, out.decode(errors="replace")
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(p.pid, sig)  # (the group this child leads: start_new_session above -- nothing of the caller's is in it)
            except ProcessLookupError:
                break
            try:
                p.wait(timeout=15)
                break
            except subprocess.TimeoutExpired:
                continue
        try:
            out, _ = p.communicate(timeout=15)  # (what it had written before it was ended)
        except Exception:  # noqa: BLE001
            out = b""
        return 124, (out or b"").decode(errors="replace")


def _last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                pass
    return None


SKETCH_ONLY = {"ont_k31": "ont_k31_sketch"}  # the same reads without the collective: the last resort of an N-rank run


def launch_ranks(args, argv=None, runner=_run_child, environ=None):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks as a CHILD `torch.distributed.run`
    (one process per GPU, rendezvous on 127.0.0.1) and relay its JSON line.  Nothing in this process has touched the GPU (no
    torch import yet), and every child is a child -- never an exec of a process that holds the device.

    The first contact with N real GPUs must not be wasted: if the child exits non-zero or prints no JSON line, a FRESH child
    runs with the next transport of the exchange -- the library's COPY transport first, then its RCCL all-to-all, then torch's
    process group as carrier (KMU_BENCH_TRANSPORT = copy | rccl | torch names where the chain starts) --, and last one with the
    sketch-only workload (no collective at all); the line that comes out says which attempt it is (`fallback_from`).  An attempt
    that is still running after KMU_BENCH_ATTEMPT_TIMEOUT seconds (default 420) is ended and counts as failed."""
    import socket
    argv = list(sys.argv[1:] if argv is None else argv)
    environ = dict(os.environ if environ is None else environ)
    # the transports of the exchange, in the order they are tried: the library's COPY transport (device copies
    # into the peers' IPC-mapped receive buffers: no kernel has to find a CU under the sketch kernels), its RCCL all-to-all, the
    # process group as carrier, and last the step without a collective.  KMU_BENCH_TRANSPORT names where to start.
    order = [("copy", "copy: the library's communicator, the all-to-all as device copies into the peers' IPC-mapped buffers"),
             ("rccl", "rccl: the library's communicator"), ("torch", "torch: the process group carries the exchange")]
    first = environ.get("KMU_BENCH_TRANSPORT", "copy")
    if environ.get("KMU_BENCH_BACKEND", "nccl") != "nccl":
        first = "torch"  # (gloo rehearsals: the ranks share devices, the process group is the only carrier)
    names = [n for n, _ in order]
    order = order[names.index(first):] if first in names else order
    attempts = [({"KMU_BENCH_TRANSPORT": n}, None, what) for n, what in order]
    if args.workload in SKETCH_ONLY:
        attempts.append(({}, SKETCH_ONLY[args.workload], "no collective: %s" % SKETCH_ONLY[args.workload]))
    failed = []
    rc = 1
    for envo, wl, what in attempts:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        av = list(argv)
        if wl:
            av = [a for i, a in enumerate(av) if a != "--workload" and (i == 0 or av[i - 1] != "--workload") and not a.startswith("--workload=")]
            av += ["--workload", wl]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + av
        env = dict(environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        env.update(envo)
        # (an attempt that hangs -- a transport that has never met this hardware -- must not take the others with it)
        try:
            rc, out = runner(cmd, env, timeout=float(environ.get("KMU_BENCH_ATTEMPT_TIMEOUT", "420")))
        except TypeError:  # (a runner without a timeout of its own: the tests' stand-ins)
            rc, out = runner(cmd, env)
        line = _last_json_line(out)
        if rc == 0 and line is not None:
            if failed:
                line["fallback_from"] = failed
            if wl:  # the line is NOT the headline workload: neither the count nor the collective ran -- the metric says so
                line["workload_substituted"] = {"asked": args.workload, "ran": wl}
                line["metric"] = "%s [SUBSTITUTED workload %s: sketch only, no count, no collective]" % (line.get("metric"), wl)
            line.setdefault("attempt", what)
            print(json.dumps(line))
            return 0
        failed.append({"attempt": what, "exit_code": rc, "json_line": line is not None})
        print("bench.py: attempt '%s' failed (exit code %d, %s JSON line)%s" % (
            what, rc, "a" if line is not None else "no", "; trying the next transport" if what != attempts[-1][2] else ""), file=sys.stderr)
    return rc or 1



# BASELINE.json's configurations next to the headline, as workloads of this script (config 2 as written: ntHash + count)
CONFIG_WORKLOADS = ("c1_super", "c2_nthash_count", "c3_k8", "c4_count", "c5_aa")


def run_configs(steps=5, warmup=3, runner=_run_child, environ=None, timeout_note=None):
    """One timed line per BASELINE configuration (one GPU's share of configs 4 and 5), each from a CHILD `python bench.py
    --workload <w>` -- a fresh process and context per workload, so that no scratch buffer, table or route decision of one run
    leaks into the next; the parent only relays: ms per step, throughput, the roofline fraction of the dominant unit, and
    whether the timed output equals the oracle's on the first 1 000 reads."""
    out = {}
    environ = dict(os.environ if environ is None else environ)
    for w in CONFIG_WORKLOADS:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", w, "--steps", str(steps), "--warmup", str(warmup),
               "--no-cpu-baseline", "--no-host-leg", "--no-configs"]
        t0 = time.time()
        rc, text = runner(cmd, environ)
        line = _last_json_line(text)
        if rc != 0 or line is None:
            out[w] = {"error": "exit code %d, %s JSON line" % (rc, "a" if line is not None else "no")}
            continue
        checks = line.get("checks") or {}
        par = [v for k, v in checks.items() if k.startswith("parity_") and isinstance(v, bool)]
        cfgl = line.get("config") or {}
        protein = "protein" in str(cfgl.get("workload", ""))
        roof = line.get("roofline") or {}
        out[w] = {"workload": cfgl.get("workload"), "ms_per_step": line.get("ms_per_step"), "value": line.get("value"),
                  "unit": "Gresidues/s" if protein else "Gbases/s", "steps": line.get("steps"), "warmup": line.get("warmup"),
                  "roofline_frac": roof.get("frac"), "roofline_kernel": roof.get("kernel"), "roofline_avg_launch_ms": roof.get("avg_launch_ms"),
                  "parity_ok": bool(par) and all(par), "parity_reads": checks.get("parity_reads"),
                  "count_conservation_ok": checks.get("count_conservation_ok"), "wall_s": round(time.time() - t0, 2)}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch
    import torch.distributed as dist
    from kmerutils_amd import _abi as A
    from kmerutils_amd import dist as kdist
    from kmerutils_amd import lib, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # KMU_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (the ranks
    # then share devices and the library's exchange is carried by the process group); the measured configuration is
    # the library's own RCCL communicator
    backend = os.environ.get("KMU_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one node: the RCCL bootstrap (torch's communicator and the library's own) over the loopback interface, whatever the
        # container's other interfaces and its hostname resolve to; the data path is xGMI / shared memory either way
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if world > 1:  # leave a few CUs to the RCCL kernels that run under the (persistent, one-workgroup-per-CU) sketch kernel
        os.environ.setdefault("KMU_PMH_RESERVE_CUS", "16")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cfg = workload_cfg(args)
    if args.scaling == "strong":
        # the job's reads split over the ranks (different reads per rank).  BASELINE.json states config 4 and config 5 as job
        # totals (50 M reads, 5 M proteins): those are what a strong-scaling line of these workloads divides; the others
        # divide the per-GPU figure the weak default gives every rank
        tot = cfg.get("strong_total_reads", cfg["n_reads"])
        cfg["total_bases"] = cfg["total_bases"] * (tot / cfg["n_reads"]) / world
        cfg["n_reads"] = max(1, tot // world)

    # ---- synthetic reads, generated in HBM (same genome on every rank, rank-specific reads) -----------------
    t_gen = time.time()
    n_reads = cfg["n_reads"]
    torch.manual_seed(cfg["seed"])
    bases, offsets, lens = _gen(synth, cfg, dev, rank)
    total_bases = int(offsets[-1].item())
    nk = int(np.maximum(lens - cfg["k"] + 1, 0).sum())
    torch.cuda.synchronize()
    torch.cuda.empty_cache()  # the generator's temporaries go back to the device: the library allocates with hipMalloc
    t_gen = time.time() - t_gen

    stream = torch.cuda.Stream(device=dev)
    ctx = lib.Context(local_rank, stream=stream.cuda_stream, async_device=True)
    transport = None
    # KMU_BENCH_FORCE_COMM=1: a communicator (RCCL, to self) also on one rank, so that one GPU executes and times the very
    # path N ranks take -- census, route choice, all-to-all, build from what arrives, finalize
    use_comm = (world > 1 or bool(os.environ.get("KMU_BENCH_FORCE_COMM"))) and cfg["count"]
    cdev = dev if backend == "nccl" else torch.device("cpu")
    comm_fallbacks = []

    def all_ok(ok):
        """did every rank get through?  (a rank that failed must not leave the others in a collective it never enters)"""
        if world == 1:
            return ok
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() == 1.0)

    if use_comm:
        # the library's own communicator, RCCL over xGMI (the id travels over the process group), probed with a small
        # distributed count before anything is timed; if any rank fails there, EVERY rank drops it and the process group
        # carries the exchange instead (KMU_BENCH_TRANSPORT=torch asks for that from the start); if that fails too the
        # step loses its collective (sketch only) -- and the line says so
        want = os.environ.get("KMU_BENCH_TRANSPORT", "copy" if backend == "nccl" else "torch")
        chain = ["copy", "rccl", "torch"] if backend == "nccl" else ["torch"]
        for tr in (chain[chain.index(want):] if want in chain else [want]):
            err = None
            try:
                if os.environ.get("KMU_BENCH_FAIL_TRANSPORT") == tr:  # (rehearsals of the fallback itself)
                    raise RuntimeError("KMU_BENCH_FAIL_TRANSPORT=%s" % tr)
                kdist.init_comm(ctx, transport=tr)
                pn = min(1000, n_reads)
                pb = int(offsets[pn].item())  # (the probe's table: room for every k-mer of its reads -- the hint counts distinct k-mers)
                pc = ctx.counter(cfg["kmer_type"], cfg["k"], 8, max(1 << 20, pb), distributed=True)
                pc.add_reads(bases[:pb], offsets[:pn + 1])
                pc.finalize()
                ctx.synchronize()
                pc.close()
            except Exception as e:  # noqa: BLE001 -- whatever it is, the other ranks must hear of it
                err = "%s: %s" % (type(e).__name__, e)
            if all_ok(err is None):
                transport = tr
                break
            comm_fallbacks.append({"transport": tr, "rank": rank, "error": err})
            try:
                ctx.comm_destroy()
            except Exception:  # noqa: BLE001
                pass
        else:
            use_comm = False
            if cfg["sketch"]:
                cfg["count"] = False  # no collective left: the sketch half of the step, sharded
    p = A.SketchParams(cfg["algo"], cfg["kmer_type"], cfg["k"], cfg["m"], cfg["sig"], cfg["hasher"], cfg["fhash"], 0,
                       A.MODE_PER_SEQ, A.INPUT_ASCII, A.MEM_DEVICE, 0)
    sig_dtype = {A.SIG_U32: torch.int32, A.SIG_U64: torch.int64, A.SIG_F32: torch.float32,
                 A.SIG_F64: torch.float64}[cfg["sig"]]
    sig = torch.zeros((n_reads, cfg["m"]), dtype=sig_dtype, device=dev) if cfg["sketch"] else None
    # what a caller knows is the number of k-mer OCCURRENCES of its reads: the first add sizes the table from the duplication it
    # measures (KMU_COUNT_HINT_OCCURRENCES; weak scaling: every rank ends up owning about one shard's worth of distinct k-mers)
    counter = ctx.counter(cfg["kmer_type"], cfg["k"], 8, max(nk, 1024), distributed=use_comm, hint_occurrences=True) if cfg["count"] else None
    nth = torch.zeros(total_bases + 64, dtype=torch.int64, device=dev) if cfg["nthash"] else None

    def step_device():
        """one pass of the hot path over the reads resident in HBM, results left in HBM"""
        if cfg["nthash"]:
            ctx.nthash(bases, offsets, cfg["k"], want_strand=False, out=nth)
        if cfg["count"]:
            counter.reset()
        if cfg["sketch"]:  # the reads once, both results; N > 1: the all-to-all is in flight under the sketch kernels
            ctx.sketch_count(bases, offsets, p, counter=counter, out=sig)
        elif cfg["count"]:
            counter.add_reads(bases, offsets)
        if cfg["count"] and use_comm:
            counter.finalize()

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed(fn, profile):
        with torch.cuda.stream(stream):
            for _ in range(args.warmup):
                fn()
            barrier()
            if profile:
                ctx.profile_reset()
                ctx.profile_enable(True)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record(stream)
            for _ in range(args.steps):
                fn()
            ev1.record(stream)
            barrier()
            t1 = time.perf_counter()
            if profile:
                ctx.profile_enable(False)
        el = t1 - t0
        if world > 1:
            cdev = dev if backend == "nccl" else torch.device("cpu")
            tt = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, ev0.elapsed_time(ev1)

    elapsed, dev_ms = timed(step_device, True)
    stats = ctx.profile_get()
    comm_stats = ctx.comm_stats() if use_comm and cfg["count"] else None
    comm_ranks = None
    if world > 1:  # what every rank measured: its exchange (the link rate the route model assumes) and its kernels
        mine_r = {"rank": rank, "kernels_ms": {n: round(ms / max(1, ln), 3) for n, (ln, ms) in stats.items() if ln},
                  "ms_per_step": elapsed / args.steps * 1e3}
        if comm_stats:
            mine_r.update({k: comm_stats[k] for k in ("exchange_ms", "bytes_sent", "bytes_received", "exchange_gbps_out",
                                                      "exchange_gbps_in", "records_local", "kmers_local", "route")})
        comm_ranks = [None] * world
        dist.all_gather_object(comm_ranks, mine_r)
    if world > 1:
        tb = torch.tensor([total_bases], dtype=torch.float64, device=cdev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        job_bases = float(tb.item())
    else:
        job_bases = float(total_bases)

    # ---- sanity of what was computed in the timed steps (not timed) ----------------------------------------
    checks = {}
    if cfg["sketch"]:
        checks["sig_checksum"] = int(sig.view(torch.int64 if sig.element_size() == 8 else torch.int32).sum().item())
    if cfg["count"]:
        occ = counter.nb_occurrences()
        dis = counter.nb_distinct()
        if world > 1:
            cdev = dev if backend == "nccl" else torch.device("cpu")
            t3 = torch.tensor([occ, dis, nk], dtype=torch.float64, device=cdev)
            dist.all_reduce(t3, op=dist.ReduceOp.SUM)
            occ, dis, nk_job = int(t3[0].item()), int(t3[1].item()), int(t3[2].item())
        else:
            nk_job = nk
        checks["nb_distinct"] = dis
        # the counts held add up to the k-mers that went in -- unless a k-mer's count has reached the table's ceiling (a homopolymer
        # in real data): then the sum is a lower bound and the line says how many k-mers that concerns
        sat = counter.nb_saturated()
        if world > 1:
            t1 = torch.tensor([sat], dtype=torch.float64, device=cdev)
            dist.all_reduce(t1, op=dist.ReduceOp.SUM)
            sat = int(t1.item())
        checks["count_saturated_kmers"] = sat
        checks["count_conservation_ok"] = bool(occ == nk_job if sat == 0 else occ <= nk_job)
    if not args.no_parity:
        # every rank checks what ITS timed steps left behind against the oracle (rows of its first 1 000 reads; counts of
        # their k-mers, filtered by owner at N > 1); the booleans are AND-ed over the ranks
        mine = parity_check(cfg, ctx, bases, offsets, sig, counter, nth, rank=rank, world=world if use_comm else 1)
        if world > 1:
            cdev = dev if backend == "nccl" else torch.device("cpu")
            keys = sorted(k for k, v in mine.items() if isinstance(v, bool))
            tb = torch.tensor([1.0 if mine[k] else 0.0 for k in keys], dtype=torch.float64, device=cdev)
            dist.all_reduce(tb, op=dist.ReduceOp.MIN)
            for k, v in zip(keys, tb.tolist()):
                mine[k] = bool(v == 1.0)
            mine["parity_ranks"] = world
        checks.update(mine)

    # ---- the same step from pinned host memory to pinned host memory (SURVEY 8d's definition of the metric) ---------
    host = None
    if cfg["sketch"] and not args.no_host_leg and not cfg.get("protein"):
        h_bases = torch.empty(total_bases, dtype=torch.uint8).pin_memory()
        h_bases.copy_(bases[:total_bases])
        h_off = offsets.cpu().pin_memory()
        h_sig = torch.zeros((n_reads, cfg["m"]), dtype=sig_dtype).pin_memory()
        ph = A.SketchParams.from_buffer_copy(p)

        def step_host():
            if cfg["count"]:
                counter.reset()
            ctx.sketch_count(h_bases, h_off, ph, counter=counter, out=h_sig)
            if cfg["count"] and use_comm:
                counter.finalize()

        h_el, _ = timed(step_host, True)
        hstats = ctx.profile_get()
        host = {"value": job_bases * args.steps / h_el / 1e9, "unit": "Gbases/s", "ms_per_step": h_el / args.steps * 1e3,
                # the library's own timers over the leg: a kernel's launches of one step (one per chunk of the upload) summed
                "kernel_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in sorted(hstats.items())},
                "what": "bases in pinned host memory -> signatures in pinned host memory + counts on the device: one chunked "
                        "upload overlapped with the sketch kernels, rows downloaded under the count build (kmu_sketch_count)",
                "bytes_up": total_bases, "bytes_down": n_reads * cfg["m"] * h_sig.element_size()}
        if cfg["sketch"]:
            same = bool(torch.equal(h_sig, sig.cpu()))
            checks["host_leg_equals_device_leg"] = same

    if rank == 0:
        value = job_bases * args.steps / elapsed / 1e9
        # ---- roofline of the dominant unit (HBM-bound integer path) ------------------------------------------
        sigw = 8 if cfg["sig"] in (A.SIG_U64, A.SIG_F64) else 4
        kern, dom, per_step = kernel_units(stats, args.steps, total_bases, nk, n_reads, cfg["m"], sigw,
                                           counter.table_bytes() if counter is not None else 0)
        roofline = None
        if dom:
            ach = kern[dom]["GBps"]
            roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(cfg, total_bases, dom, per_step),
                        "traffic_source": "profiles/pmc_latest.json: the builder's rocprofv3 --pmc passes of this workload and size (FETCH_SIZE / "
                                          "WRITE_SIZE, corrected as the guide prescribes), committed with the code -- NOT measured in this run",
                        "avg_launch_ms": kern[dom]["avg_ms"], "alg_bytes_per_launch": kern[dom]["alg_bytes"],
                        "what": "SURVEY 8(d) bytes of the unit (sketch: bases in + signature rows out; count: bases in + 16 B per "
                                "k-mer occurrence) / the summed launch time of the unit's kernels in one step"}
            # the other units of the step, priced the same way
            # (every unit also from the instruction-issue side: the sketch unit is far from HBM because it is bound there)
            roofline["units"] = {n: {"frac": e["GBps"] / HBM_PEAK_GBS, "avg_launch_ms": e["avg_ms"], "alg_bytes_per_launch": e["alg_bytes"],
                                     "traffic": pmc_traffic(cfg, total_bases, n, e.get("per_step")),
                                     "alu": pmc_alu(cfg, total_bases, n, e["avg_ms"], e.get("per_step"))}
                                 for n, e in kern.items() if "alg_bytes" in e}
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(cfg, bases, offsets, lens, args.cpu_sample_reads, args.cpu_threads)
        configs = None
        if world == 1 and cfg["name"] == "ont_k31" and not args.no_configs and not (args.reads or args.bases or args.genome):
            # the children get the whole card: this process first gives its device memory back (a child that finds the card half
            # full takes other routes -- config 3's list-emitting route wants 52 GB of scratch -- and would not be the workload)
            if counter is not None:
                counter.close()
                counter = None
            ctx.close()
            del bases, offsets, sig, nth
            if host is not None:
                del h_bases, h_sig, h_off
            torch.cuda.empty_cache()
            configs = run_configs()
        out = {
            "metric": "Gbases/sec k-mer+sketch throughput, k=31, 200 sketches/read",
            "value": value, "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": _describe(cfg, n_reads, total_bases), "workload_name": cfg["name"], "reads_per_gpu": n_reads,
                       "bases_per_gpu": total_bases, "kmers_per_gpu": nk, "k": cfg["k"], "sketch_size": cfg["m"],
                       "sketch": cfg["sketch"], "count": cfg["count"], "parallelism": "reads sharded x%d" % world,
                       "residency": "value: reads and results resident in HBM; host_to_host: pinned host memory in and out"},
            "roofline": roofline, "alu": pmc_alu(cfg, total_bases, dom, kern[dom]["avg_ms"], per_step) if dom else None,
            "cpu_baseline": cpu, "host_to_host": host, "value_device_resident": value,
            "value_host_to_host": host["value"] if host else None,
            # SURVEY.md 8(d) defines the metric host to host (bases in host memory -> signatures in host memory + counts on the
            # device); `value` is the bench contract's device-resident figure.  Quote THIS one as the 8(d) number.
            "metric_value_8d": host["value"] if host else None,
            "kernels": kern, "device_ms_per_step": dev_ms / args.steps,
            # for whoever samples the card from outside (a sampler that saw 0 % busy): the launches the library timed
            # inside the timed region, their summed device time (hipEvents), and the region's wall time
            "gpu_work": {"timed_kernel_launches": int(sum(ln for ln, _ in stats.values())), "timed_kernel_ms": float(sum(ms for _, ms in stats.values())),
                         "timed_region_wall_ms": elapsed * 1e3, "timed_region_device_ms": dev_ms},
            "checks": checks, "comm": _comm_summary(comm_stats, transport, comm_fallbacks, comm_ranks), "gen_seconds": t_gen,
            # the other BASELINE configurations on this GPU (configs 4 / 5: one GPU's shard), each timed by a child of this run
            "configs": configs,
        }
        if comm_fallbacks:
            out["fallback_from"] = comm_fallbacks
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _comm_summary(comm_stats, transport, fallbacks, ranks):
    """the `comm` object of the line: rank 0's statistics of the last step, and over all ranks the measured exchange (min /
    mean / max) with every rank's kernel times -- the first run on N real GPUs calibrates KMU_XGMI_GBPS from these"""
    if not comm_stats and not ranks:
        return None
    out = dict(comm_stats or {}, transport=transport)
    if fallbacks:
        out["fallback_from"] = fallbacks
    if ranks:
        for key in ("exchange_ms", "exchange_gbps_out", "exchange_gbps_in", "bytes_sent", "ms_per_step"):
            v = [r[key] for r in ranks if r and key in r]
            if v:
                out[key + "_ranks"] = {"min": min(v), "mean": sum(v) / len(v), "max": max(v)}
        out["per_rank"] = ranks
    return out


# the kernels of the two units of a step, by the names the library's timers carry (= the kernels' function names, the
# names rocprofv3 prints)
SKETCH_UNIT = ("k_multiset_uq", "k_multiset_short", "k_sketch_smallk", "k_sketch_pmh3a", "k_sketch_pmh3a_redo", "k_pmh_points",
               "k_pmh_points_short")
COUNT_UNIT = ("k_part_hist1", "k_part_scan1", "k_part_scatter1", "k_arr_hist", "k_arr_scan", "k_arr_scatter", "k_part_build",
              "k_part_build_q", "k_count_add_spill", "k_count_add_flat",
              # N > 1, minimizer owners: the sender's census + record scatter, the receiver's first level from records
              "k_smer_census", "k_smer_scan", "k_smer_scatter", "k_smer_scatter1", "k_smer_expand", "k_sample_distinct")


def kernel_units(stats, steps, total_bases, nk, n_reads, m, sigw, table_bytes):
    """Per-kernel launch statistics of the timed steps -> (kernels, dominant unit, launches per step of its parts).

    `alg_bytes` are SURVEY.md 8(d)'s ALGORITHMIC bytes and exist for whole units only: the sketch unit (every kernel between
    bases and signature rows: bases in + rows out) and the count unit (every kernel between bases and the table: bases in +
    16 B per k-mer occurrence).  A unit's time is the sum of its kernels' launches in one step.  Single kernels of a unit
    carry `design_bytes` instead -- what THIS design makes the kernel move (partition streams, the table image): a
    bandwidth figure of the kernel, not a roofline figure of the path."""
    sketch_alg = total_bases + n_reads * m * sigw
    count_alg = total_bases + nk * 16
    design = {
        "k_pmh_points": nk * 12 + n_reads * m * sigw,         # (key, weight) lists in, rows out
        "k_part_hist1": total_bases,
        "k_part_scatter1": total_bases + nk * 8,
        "k_arr_hist": nk * 8,
        "k_arr_scatter": nk * 16,
        "k_part_build": nk * 8 + table_bytes,                  # leaves in, the table image out (12 bytes per slot)
        "k_part_build_q": nk * 8 + table_bytes,                # ... (8 bytes per slot)
    }
    single = {"k_sketch_super": sketch_alg, "k_oph_reads": sketch_alg, "k_nthash": total_bases + nk * 8}
    kern = {}
    for name, (launches, ms) in stats.items():
        if launches:
            avg = ms / launches
            ent = {"launches": launches, "avg_ms": avg}
            if name in design:
                ent["design_bytes"] = design[name]
                ent["design_GBps"] = design[name] / (avg * 1e-3) / 1e9
            if name in single:
                ent["alg_bytes"] = single[name]
                ent["GBps"] = single[name] / (avg * 1e-3) / 1e9
            kern[name] = ent
    for unit, alg in ((SKETCH_UNIT, sketch_alg), (COUNT_UNIT, count_alg)):
        parts = [n for n in unit if n in kern]
        if not parts:
            continue
        ms = sum(kern[n]["avg_ms"] * kern[n]["launches"] for n in parts) / steps  # the unit's kernel time in one step
        kern["+".join(parts)] = {"launches": steps, "avg_ms": ms, "alg_bytes": alg, "GBps": alg / (ms * 1e-3) / 1e9,
                                 "per_step": {n: kern[n]["launches"] / steps for n in parts}}
    cand = [n for n in kern if "alg_bytes" in kern[n]]
    dom = max(cand, key=lambda n: kern[n]["avg_ms"] * kern[n]["launches"], default=None)
    return kern, dom, (kern[dom].get("per_step") or {dom: kern[dom]["launches"] / steps}) if dom else {}


def _np_int64_hash(key):
    """Thomas Wang's hash64shift on a numpy uint64 array: DispatchableT::dispatch of Kmer64bit, kmercount.rs:412-420"""
    with np.errstate(over="ignore"):
        key = key.astype(np.uint64)
        key = ~key + (key << np.uint64(21))
        key = key ^ (key >> np.uint64(24))
        key = (key + (key << np.uint64(3))) + (key << np.uint64(8))
        key = key ^ (key >> np.uint64(14))
        key = (key + (key << np.uint64(2))) + (key << np.uint64(4))
        key = key ^ (key >> np.uint64(28))
        key = key + (key << np.uint64(31))
    return key


def _np_int32_hash(key):
    with np.errstate(over="ignore"):
        key = key.astype(np.uint32)
        key = ~key + (key << np.uint32(15))
        key = key ^ (key >> np.uint32(12))
        key = key + (key << np.uint32(2))
        key = key ^ (key >> np.uint32(4))
        key = key * np.uint32(2057)
        key = key ^ (key >> np.uint32(16))
    return key


def parity_check(cfg, ctx, bases, offsets, sig, counter, nth, n_check=1000, rank=0, world=1):
    """SURVEY.md 8(d) parity set: the first 1 000 reads of the workload.  The rows / counts / hashes the TIMED steps left
    behind are compared with the oracle's on those reads (the oracle is the checker here, nothing it computes is timed).
    world > 1 (a distributed counter): after kmu_count_finalize this rank holds exactly the k-mers with owner == rank (the
    owner of the k-mer's minimizer by default, int64_hash(kmer) % world -- kmercount.rs:412-420 -- with KMU_COUNT_OWNER=hash):
    the k-mers of its 1 000 reads that it owns must be there with at least the oracle's count, the ones it does not own
    must be absent."""
    import torch
    from kmerutils_amd import _abi as A
    from oracle import oracle as O
    n = min(n_check, len(offsets) - 1)
    nb = int(offsets[n].item())
    hb = bases[:nb].cpu().numpy()
    ho = offsets[:n + 1].cpu().numpy().astype(np.uint64)
    out = {"parity_reads": n}
    if cfg["sketch"]:
        p = A.SketchParams(cfg["algo"], cfg["kmer_type"], cfg["k"], cfg["m"], cfg["sig"], cfg["hasher"], cfg["fhash"], 0, 0, 0, 0, 0)
        want = O.sketch(hb, ho, p)
        ctx.synchronize()
        got = sig[:n].cpu().numpy()
        out["parity_rows_ok"] = bool(np.array_equal(got.view(np.uint8), np.ascontiguousarray(want).view(np.uint8)))
    if cfg["count"]:
        # every canonical k-mer of those reads: the table's count is at least the oracle's count over the 1 000 reads, and
        # equal for a k-mer the full read set holds nowhere else -- checked through count conservation + this lower bound
        oc = O.Counter(cfg["kmer_type"], cfg["k"], 16, 1 << 20)
        oc.add_reads(hb, ho)
        wk, wc = oc.dump(1)
        got_d = counter.query(torch.from_numpy(wk.view(np.int64)).to(bases.device))
        ctx.synchronize()  # (an asynchronous context: the query runs on the library's stream, not on torch's current one)
        got = got_d.cpu().numpy().astype(np.int64)
        if world > 1:
            w32 = A.kmer_val_bytes(cfg["kmer_type"]) == 4
            if counter.owner_kind == A.OWNER_MINIMIZER:  # the default between GPUs: the k-mer's minimizer decides (kmu_smer.h)
                own = O.minimizer_owners(wk, cfg["k"], world) == rank
            else:
                own = ((_np_int32_hash(wk).astype(np.uint64) if w32 else _np_int64_hash(wk)) % np.uint64(world)) == np.uint64(rank)
            out["parity_counts_ok"] = bool((got[own] >= np.minimum(wc[own].astype(np.int64), 255)).all() and (got[own] >= 1).all()
                                           and (got[~own] == 0).all())
            out["parity_kmers_owned"] = int(own.sum())
        else:
            out["parity_counts_ok"] = bool((got >= np.minimum(wc.astype(np.int64), 255)).all() and (got >= 1).all())
    if cfg.get("nthash"):
        ctx.synchronize()
        wh, _ = O.nthash(hb, ho, cfg["k"])
        out["parity_nthash_ok"] = bool(np.array_equal(nth[:nb].cpu().numpy().view(np.uint64), wh[:, 0]))
    return out


# wave-instructions / s the chip can issue, MEASURED (scripts/micro/valu_issue.hip, profiles/r03_valu_issue.txt): 1.03e9 per SIMD
# for full-rate instructions (v_add_u32, v_xor_b32: one every ~2.3 cycles at the 2.2-2.3 GHz the chip holds under this load,
# from two waves per SIMD up -- the guide's 2-cycle issue, MI355X_MICROARCH.md:54,473) x 4 SIMDs x 256 CUs.  The 64-bit shifts,
# 32-bit multiplies, v_mad_u64_u32, carry adds and f64 FMAs that SplitMix64 / xoshiro / the Wang hash / Exp01 consist of issue at
# HALF that rate (0.53-0.58e9 per SIMD) at any occupancy: VALU_HALF_RATE_PEAK is the ceiling of a kernel made of those alone.
VALU_ISSUE_PEAK = 1.03e9 * 4 * 256
VALU_HALF_RATE_PEAK = 0.57e9 * 4 * 256


def pmc_alu(cfg, total_bases, kernel, avg_ms, per_step=None):
    """Instruction-issue view of an ALU-bound kernel, next to the HBM roofline the tier asks for: vector wave-instructions
    per launch from the committed PMC pass (SQ_INSTS_VALU) / the launch time measured in this run, against the chip's
    full-rate VALU issue peak.  null when the committed profile is not of this workload and size.  `per_step`: launches
    per step of every kernel name of a composite unit (a name's per-launch figure is the mean over its launches)."""
    per_step = per_step or {}
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        d = json.load(open(path))
        if d.get("workload") == cfg["name"] and abs(d.get("bases_per_gpu", 0) - total_bases) < 1e-3 * total_bases:
            ks = [d["kernels"].get(n, {}) for n in kernel.split("+")]
            mult = [per_step.get(n, 1.0) for n in kernel.split("+")]
            valu = sum(k.get("SQ_INSTS_VALU_per_launch", 0) * m for k, m in zip(ks, mult)) if all("SQ_INSTS_VALU_per_launch" in k for k in ks) else None
            if valu and avg_ms:
                ach = valu / (avg_ms * 1e-3)
                return {"bound": "valu-issue", "valu_wave_insts": valu,
                        "salu_wave_insts": sum(k.get("SQ_INSTS_SALU_per_launch", 0) * m for k, m in zip(ks, mult)),
                        "achieved": ach, "peak": VALU_ISSUE_PEAK, "unit": "wave-inst/s", "frac": ach / VALU_ISSUE_PEAK,
                        "peak_half_rate_insts": VALU_HALF_RATE_PEAK, "source": "profiles/r03_valu_issue.txt"}
    except Exception:
        pass
    return None


def pmc_traffic(cfg, total_bases, kernel, per_step=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, corrected
    as MI355X_MICROARCH.md prescribes; scripts/summarize_prof.py).  PMC counters cannot be read from inside the timed
    run, so the figure is only reported when the committed profile is of this exact workload and size; else null."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        d = json.load(open(path))
        if d.get("workload") == cfg["name"] and abs(d.get("bases_per_gpu", 0) - total_bases) < 1e-3 * total_bases:
            ks = [d["kernels"].get(n, {}).get("hbm_bytes_per_launch") for n in kernel.split("+")]
            mult = [(per_step or {}).get(n, 1.0) for n in kernel.split("+")]
            return sum(v * m for v, m in zip(ks, mult)) if all(v is not None for v in ks) else None
    except Exception:
        pass
    return None


def _gen(synth, cfg, dev, rank):
    import torch
    if cfg.get("protein"):  # log-normal lengths (median 300), residues iid with UniProt-like frequencies
        res, off = synth.protein_seqs(cfg["n_reads"], cfg["seed"] * 1000 + rank)
        return torch.from_numpy(res.copy()).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev), np.diff(off.astype(np.int64))
    # same genome on every rank (genome seed), rank-specific read sampling
    bases, offsets, lens = synth.ont_reads_device(cfg["n_reads"], cfg["total_bases"], cfg["genome"],
                                                  cfg["seed"], dev, errors=cfg["errors"], fixed_len=cfg["fixed_len"],
                                                  read_seed=cfg["seed"] * 1000 + rank)
    return bases, offsets, lens


def _describe(cfg, n_reads, total_bases):
    ops = []
    if cfg["sketch"]:
        ops.append("%s m=%d" % ("ProbMinHash3a" if cfg["algo"] == 0 else "SuperMinHash", cfg["m"]))
    if cfg.get("nthash"):
        ops.append("canonical ntHash per position")
    if cfg["count"]:
        ops.append("kmercount 8-bit")
    shape = "%d protein sequences" % n_reads if cfg.get("protein") else \
        "%d x %d bp reads" % (n_reads, cfg["fixed_len"]) if cfg["fixed_len"] else \
        "%d ONT-shaped reads (log-normal lengths, 8%% errors)" % n_reads
    return "%s: %s, %.3g bases, k=%d, %s" % (cfg["name"], shape, total_bases, cfg["k"], " + ".join(ops))


def cpu_baseline(cfg, bases, offsets, lens, sample_reads, threads=0):
    """The oracle (a C restatement of the reference algorithm = `kind: port`) timed on this box's host cores over a
    bounded sample of the same workload.  Like the reference's rayon map (seqsketchjaccard.rs:245-248) the reads are
    sharded over T threads for sketching; counting uses one independent counter per thread over its shard
    (count_kmer_thread_independant, kmercount.rs:797).  ctypes releases the GIL, so the threads run in parallel."""
    from concurrent.futures import ThreadPoolExecutor
    from kmerutils_amd import _abi as A
    from oracle import oracle as O
    O.lib()
    T = threads or max(1, os.cpu_count() or 1)  # what rayon would use: every hardware thread of the box
    if not sample_reads:
        # roughly 10-30 s of wall time (measured on the 256-thread host of the GPU box: the count leg, one exact hash
        # table per thread, takes 22 s per Gbase there; the sketch leg 3 s)
        target_bases = min(T, 64) * (1.2e7 if cfg["count"] else 6e7)
        csum = np.cumsum(lens)
        sample_reads = int(min(len(lens), max(16, np.searchsorted(csum, target_bases) + 1)))
    T = max(1, min(T, sample_reads))
    nb = int(offsets[sample_reads].item())
    hb = bases[:nb].cpu().numpy()
    ho = offsets[:sample_reads + 1].cpu().numpy().astype(np.uint64)
    p = A.SketchParams(cfg["algo"], cfg["kmer_type"], cfg["k"], cfg["m"], cfg["sig"], cfg["hasher"], cfg["fhash"], 0, 0,
                       0, 0, 0)
    from kmerutils_amd.dist import shard_reads_by_bases
    shards = shard_reads_by_bases(np.diff(ho.astype(np.int64)), T)

    def shard_arrays(r0, r1):
        b0, b1 = int(ho[r0]), int(ho[r1])
        return np.ascontiguousarray(hb[b0:b1]), np.ascontiguousarray(ho[r0:r1 + 1] - ho[r0])

    parts = [shard_arrays(r0, r1) for r0, r1 in shards if r1 > r0]

    def run_sketch(part):
        t = time.perf_counter()
        O.sketch(part[0], part[1], p)
        return time.perf_counter() - t

    def run_count(part):
        t = time.perf_counter()
        c = O.Counter(cfg["kmer_type"], cfg["k"], 8, max(1024, part[0].size))
        c.add_reads(part[0], part[1])
        del c
        return time.perf_counter() - t

    t_sk = t_ct = 0.0
    one = []
    with ThreadPoolExecutor(max_workers=len(parts)) as ex:
        if cfg["sketch"]:
            t0 = time.perf_counter()
            one.append(list(ex.map(run_sketch, parts)))
            t_sk = time.perf_counter() - t0
        if cfg["count"]:
            t0 = time.perf_counter()
            one.append(list(ex.map(run_count, parts)))
            t_ct = time.perf_counter() - t0
    tot = t_sk + t_ct
    per_thread = sum(np.array(x) for x in one)  # seconds each thread spent on its shard
    rate1 = float(np.mean([parts[i][0].size / per_thread[i] for i in range(len(parts))])) / 1e9
    return {"value": nb / tot / 1e9, "unit": "Gbases/s", "cores": len(parts), "kind": "port",
            "sample": "first %d reads (%d bases) of the same read set, sharded over %d threads; sketch %.2f s + count "
                      "%.2f s wall; oracle C restatement, gcc -O2; one thread alone: %.4f Gbases/s; sketch: reads sharded like "
                      "the reference's rayon map (seqsketchjaccard.rs:245-248); count: one private exact table per thread "
                      "over its own shard, not merged -- cheaper than the reference's count_kmer_thread_independant "
                      "(kmercount.rs:797-867, where every thread scans ALL reads and keeps its key range), so this "
                      "baseline flatters the CPU"
                      % (sample_reads, nb, len(parts), t_sk, t_ct, rate1),
            "host_cpus": os.cpu_count(), "physical_cores": _physical_cores()}


def _physical_cores():
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except OSError:
        return None


if __name__ == "__main__":
    main()
