#!/usr/bin/env python3
"""diagnostics: the two device-side halves of the multi-GPU counting step, timed on one GPU at the bench size:
(1) grouping the local k-mers by owner rank (kmu_count_extract_by_owner, world = N), (2) building the owned table
from a received k-mer array of the same size (kmu_count_add_kmers)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kmerutils_amd import _abi as A, lib, synth

N = int(os.environ.get("WORLD", 8))
dev = torch.device("cuda:0")
bases, offsets, lens = synth.ont_reads_device(746_333, 4.38e9, 100_000_000, 0xC3, dev)
nk = int(np.maximum(lens - 31 + 1, 0).sum())
ctx = lib.Context(0)
ctx.profile_enable(True)
c = ctx.counter(A.KMER64BIT, 31, 8, max(nk, 1024))
res = {}
for it in range(2):
    ctx.profile_reset()
    t0 = time.perf_counter()
    kmers, bounds = c.extract_by_owner(bases, offsets, N)
    ctx.synchronize()
    res["extract_ms"] = (time.perf_counter() - t0) * 1e3
    res["extract_kernels"] = {k: round(v[1] / max(v[0], 1), 2) for k, v in ctx.profile_get().items()}
    recv = torch.as_tensor(kmers).clone()      # stands for the all-to-all's receive buffer
    torch.cuda.synchronize()
    c.reset()
    ctx.profile_reset()
    t0 = time.perf_counter()
    c.add_kmers(recv)
    ctx.synchronize()
    res["build_ms"] = (time.perf_counter() - t0) * 1e3
    res["build_kernels"] = {k: round(v[1] / max(v[0], 1), 2) for k, v in ctx.profile_get().items()}
    del recv
res["owner_sizes"] = np.diff(bounds.astype(np.int64)).tolist()
print(json.dumps(res))
