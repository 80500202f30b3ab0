"""diagnostics: compact vs open image of the partitioned build at a size with many regions per workgroup"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kmerutils_amd import _abi as A, lib, synth
n = int(os.environ.get("DBG_READS", "40000"))
dev = torch.device("cuda", 0)
bases, off, lens = synth.ont_reads_device(n, n * 5800, 100_000_000, 0xC3, dev)
nk = int(np.maximum(lens - 30, 0).sum())
res = {}
for mode in ("0", "1"):
    os.environ["KMU_COUNT_COMPACT"] = mode
    ctx = lib.Context(0)
    c = ctx.counter(A.KMER64BIT, 31, 8, nk)
    c.add_reads(bases, off)
    st = (c.nb_distinct(), c.nb_unique(), c.nb_occurrences())
    hp = ctx.kmer_hashes(bases, off, A.KMER64BIT, 31, A.FHASH_CANON_VALUE)
    q = c.query(hp[: 50_000_000].contiguous())
    st2 = (c.nb_distinct(), c.nb_unique(), c.nb_occurrences())
    res[mode] = (st, st2, q.cpu())
    print(mode, st, st2, int((q == 0).sum()), flush=True)
    c.close(); ctx.close()
print("equal queries:", bool(torch.equal(res["0"][2], res["1"][2])), "stats", res["0"][0] == res["1"][0], res["0"][1] == res["1"][1])
