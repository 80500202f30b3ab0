#!/usr/bin/env python3
"""diagnostics: per-pack cost of kmu_sketch (k = 8, m = 200) on packs of 10 000 ONT-shaped reads taken as ranges of one device array"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from kmerutils_amd import _abi as A, lib, synth
dev = torch.device("cuda:0")
bases, offsets, lens = synth.ont_reads_device(100_000, 6e8, 100_000_000, 0xC3, dev)
ctx = lib.Context(0)
p = A.SketchParams(A.ALGO_PROB3A, A.KMER32BIT, 8, 200, A.SIG_U32, 0, A.FHASH_CANON_INVHASH, 0, 0, A.INPUT_ASCII, A.MEM_DEVICE, 0)
ctx.profile_enable(True)
sig = torch.zeros((10000, 200), dtype=torch.int32, device=dev)
for first in range(0, 100_000, 10_000):
    o = offsets[first:first + 10_001]
    ctx.profile_reset()
    t0 = time.perf_counter()
    ctx.sketch(bases, o, p, out=sig)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    st = ctx.profile_get()
    print("pack %6d: wall %.3f ms, kernels %s, max len %d, bases %d" % (first, dt * 1e3, {k: round(v[1], 3) for k, v in st.items()},
          int(lens[first:first + 10000].max()), int(lens[first:first + 10000].sum())))
