"""diagnostics: the sketch (vector-ALU bound) and the count build (HBM bound) of the headline workload on DISJOINT CU sets
at the same time -- two contexts on two CU-masked streams (hipExtStreamCreateWithCUMask), one host thread each.
Prints the wall time of the pair for a few splits next to the back-to-back time on the whole chip.
usage (GPU box): python scripts/dbg_overlap.py [count_cus ...]"""
import ctypes as C
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

hip = C.CDLL("libamdhip64.so")


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) the stream may use"""
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s.value


def main():
    splits = [int(x) for x in sys.argv[1:]] or [96, 128]
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
    torch.cuda.synchronize()

    def run(mode, count_cus, layout):
        all_cus = list(range(256))
        if mode == "seq":
            sa = sb = None
        else:
            if layout == "low":      # the first count_cus mask bits
                cb = [i for i in all_cus if i < count_cus]
            else:                    # whole groups of the (bit % 8) classes
                cb = [i for i in all_cus if (i % 8) < count_cus // 32]
            sk = [i for i in all_cus if i not in set(cb)]
            sa, sb = masked_stream(sk), masked_stream(cb)
        ts = torch.cuda.Stream(device=dev)
        ctx_s = lib.Context(0, stream=sa if sa else ts.cuda_stream, async_device=True)
        ctx_c = lib.Context(0, stream=sb if sb else ts.cuda_stream, async_device=True)
        counter = ctx_c.counter(cfg["kmer_type"], cfg["k"], 8, max(nk, 1024))
        if mode != "seq":
            os.environ["KMU_PMH_RESERVE_CUS"] = str(count_cus)
        else:
            os.environ.pop("KMU_PMH_RESERVE_CUS", None)

        def do_sketch():
            ctx_s.sketch_count(bases, offsets, p, counter=None, out=sig)
            ctx_s.synchronize()

        def do_count():
            counter.reset()
            counter.add_reads(bases, offsets)
            ctx_c.synchronize()

        res = []
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "seq":
                do_sketch()
                t1 = time.perf_counter()
                do_count()
                t2 = time.perf_counter()
                res.append((t2 - t0, t1 - t0, t2 - t1))
            else:
                tt = {}

                def wrap(name, fn):
                    a = time.perf_counter()
                    fn()
                    tt[name] = time.perf_counter() - a
                th = [threading.Thread(target=wrap, args=("s", do_sketch)), threading.Thread(target=wrap, args=("c", do_count))]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                res.append((time.perf_counter() - t0, tt["s"], tt["c"]))
        chk = int(sig.sum().item())
        nd = counter.nb_distinct()
        print("%s count_cus=%s layout=%s: pair / sketch / count ms = %s  (sig sum %d, distinct %d)" % (
            mode, count_cus, layout, ["%.1f / %.1f / %.1f" % tuple(1e3 * x for x in r) for r in res], chk, nd), flush=True)
        counter.close()
        ctx_s.close()
        ctx_c.close()
        torch.cuda.empty_cache()

    run("seq", 0, "-")
    for c in splits:
        for layout in ("low", "mod8"):
            run("par", c, layout)


if __name__ == "__main__":
    main()
