"""diagnostics: a big table filled in batches (a read set counted chunk by chunk): direct insertion against the partitioned build on
the occupied table, per batch size.  usage (GPU box): python scripts/r03_incr.py"""
import os
import sys
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
        workload = "ont_k31_count"
        reads = 0
        bases = 0.0
        genome = 0
        sketch_size = 0
    cfg = bench.workload_cfg(Args)
    torch.manual_seed(cfg["seed"])
    bases, offsets, lens = bench._gen(synth, cfg, dev, 0)
    nk = int(np.maximum(lens - cfg["k"] + 1, 0).sum())
    n = cfg["n_reads"]
    ctx = lib.Context(0)
    for parts in (4, 8, 16, 32):
        cuts = [int(i * n / parts) for i in range(parts + 1)]
        for path in ("direct", "partitioned", "auto"):
            if path == "auto":
                os.environ.pop("KMU_COUNT_PATH", None)
            else:
                os.environ["KMU_COUNT_PATH"] = path
            c = ctx.counter(cfg["kmer_type"], cfg["k"], 8, max(nk, 1024))
            torch.cuda.synchronize()
            ts = []
            for i in range(parts):
                a, b = cuts[i], cuts[i + 1]
                sub_off = offsets[a:b + 1]
                t0 = time.perf_counter()
                c.add_reads(bases, sub_off)
                ctx.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
            nd = c.nb_distinct()
            print("1/%d of the batch per add, %s: first %.1f ms, later mean %.1f ms, total %.1f ms (distinct %d)" % (
                parts, path, ts[0], float(np.mean(ts[1:])), sum(ts), nd), flush=True)
            c.close()
    os.environ.pop("KMU_COUNT_PATH", None)


if __name__ == "__main__":
    main()
