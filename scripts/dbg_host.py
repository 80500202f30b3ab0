"""diagnostics: kernel launches and times of ONE host-to-host kmu_sketch_count at bench size"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kmerutils_amd import _abi as A, lib, synth
dev = torch.device("cuda", 0)
n = int(os.environ.get("DBG_READS", "746333"))
bases, off, lens = synth.ont_reads_device(n, n * 5868.4, 100_000_000, 0xC3, dev)
tb = int(off[-1].item()); nk = int(np.maximum(lens - 30, 0).sum())
hb = torch.empty(tb, dtype=torch.uint8).pin_memory(); hb.copy_(bases[:tb]); ho = off.cpu().pin_memory()
sig = torch.zeros((n, 200), dtype=torch.int64).pin_memory()
del bases; torch.cuda.empty_cache()
ctx = lib.Context(0)
c = ctx.counter(A.KMER64BIT, 31, 8, nk)
p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 200, A.SIG_U64, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
for it in range(3):
    c.reset(); ctx.profile_reset(); ctx.profile_enable(it == 2)
    t = time.perf_counter(); ctx.sketch_count(hb, ho, p, counter=c, out=sig); dt = time.perf_counter() - t
    print("step", it, "%.1f ms" % (dt * 1e3), flush=True)
ctx.profile_enable(False)
for k, (nl, ms) in sorted(ctx.profile_get().items()): print("%-22s %3d launches %8.2f ms total" % (k, nl, ms))
