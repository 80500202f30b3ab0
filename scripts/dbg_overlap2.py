"""diagnostics (round 3): the sketch and the count of the headline workload AT THE SAME TIME on two plain streams (no CU masks):
two contexts, one host thread each; the wall time of the pair next to the back-to-back time.  The count kernels of round 3 are
closer to the memory system's rate than round 2's (dbg_overlap.py), the points kernel is bound by instruction issue.
usage (GPU box): python scripts/dbg_overlap2.py"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kmerutils_amd import _abi as A  # noqa: E402
from kmerutils_amd import lib, synth  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)

    class Args:
        workload = "ont_k31"
        reads = 0
        bases = 0.0
        genome = 0
        sketch_size = 0
    cfg = bench.workload_cfg(Args)
    torch.manual_seed(cfg["seed"])
    bases, offsets, lens = bench._gen(synth, cfg, dev, 0)
    nk = int(np.maximum(lens - cfg["k"] + 1, 0).sum())
    n_reads = cfg["n_reads"]
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    p = A.SketchParams(cfg["algo"], cfg["kmer_type"], cfg["k"], cfg["m"], cfg["sig"], cfg["hasher"], cfg["fhash"], 0,
                       A.MODE_PER_SEQ, A.INPUT_ASCII, A.MEM_DEVICE, 0)
    sig = torch.zeros((n_reads, cfg["m"]), dtype=torch.int64, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ctx_s = lib.Context(0, stream=s1.cuda_stream, async_device=True)
    ctx_c = lib.Context(0, stream=s2.cuda_stream, async_device=True)
    counter = ctx_c.counter(cfg["kmer_type"], cfg["k"], 8, max(nk, 1024))

    def do_sketch(delay=0.0):
        if delay:
            time.sleep(delay)
        ctx_s.sketch_count(bases, offsets, p, counter=None, out=sig)
        ctx_s.synchronize()

    def do_count(delay=0.0):
        if delay:
            time.sleep(delay)
        counter.reset()
        counter.add_reads(bases, offsets)
        ctx_c.synchronize()

    def pair(ds, dc):
        tt = {}

        def wrap(name, fn, d):
            a = time.perf_counter()
            fn(d)
            tt[name] = time.perf_counter() - a
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=wrap, args=("s", do_sketch, ds)), threading.Thread(target=wrap, args=("c", do_count, dc))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return 1e3 * (time.perf_counter() - t0), 1e3 * tt["s"], 1e3 * tt["c"]

    for it in range(2):  # warm-up (buffers)
        do_sketch()
        do_count()
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        do_sketch()
        t1 = time.perf_counter()
        do_count()
        t2 = time.perf_counter()
        print("back to back: pair %.1f ms = sketch %.1f + count %.1f" % (1e3 * (t2 - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
    chk0, nd0 = int(sig.sum().item()), counter.nb_distinct()
    for ds, dc in ((0, 0), (0, 0), (0, 0), (0, 0.015), (0, 0.030), (0.015, 0), (0.030, 0)):
        r = pair(ds, dc)
        print("together (sketch starts +%.0f ms, count +%.0f ms): pair %.1f ms, sketch %.1f, count %.1f  (rows %s, table %s)" % (
            1e3 * ds, 1e3 * dc, r[0], r[1], r[2], int(sig.sum().item()) == chk0, counter.nb_distinct() == nd0), flush=True)


if __name__ == "__main__":
    main()
