#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload: reads in HOST memory (pinned, and ordinary pageable), kmu_sketch +
kmu_count_add_reads in KMU_MEM_HOST mode (the library stages H2D / D2H itself), signatures back in host memory.
Reported in DESIGN.md next to the HBM-resident `value` of bench.py; never the bench value."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kmerutils_amd import _abi as A
from kmerutils_amd import lib, synth

dev = torch.device("cuda:0")
n_reads, total = int(os.environ.get("READS", 746_333)), float(os.environ.get("BASES", 4.38e9))
bases, offsets, lens = synth.ont_reads_device(n_reads, total, 100_000_000, 0xC3, dev)
nb = int(offsets[-1].item())
nk = int(np.maximum(lens - 31 + 1, 0).sum())
hb_pinned = torch.empty(nb + 64, dtype=torch.uint8).pin_memory()
hb_pinned[:nb].copy_(bases[:nb])
off_np = offsets.cpu().numpy().astype(np.uint64)
del bases
torch.cuda.synchronize()
ctx = lib.Context(0)
p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 200, A.SIG_U64, 0, A.FHASH_CANON_INVHASH, 0, 0, A.INPUT_ASCII, A.MEM_HOST, 0)
counter = ctx.counter(A.KMER64BIT, 31, 8, max(nk, 1024))
sig_pinned = torch.empty((n_reads, 200), dtype=torch.int64).pin_memory()
out = {}
for name, hb, sig in (("pinned", hb_pinned.numpy(), sig_pinned.numpy().view(np.uint64)),
                      ("pageable", np.array(hb_pinned.numpy()), np.zeros((n_reads, 200), np.uint64))):
    res = {}
    for it in range(2):
        t0 = time.perf_counter()
        ctx.sketch(hb, off_np, p, out=sig)
        t1 = time.perf_counter()
        counter.reset()
        counter.add_reads(hb, off_np)
        ctx.synchronize()
        t2 = time.perf_counter()
        res = {"sketch_s": t1 - t0, "count_s": t2 - t1, "Gbases_per_s": nb / (t2 - t0) / 1e9}
    out[name] = res
print(json.dumps({"bases": nb, "reads": n_reads, **out}))
