import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kmerutils_amd import _abi as A, lib, synth
from oracle import oracle as O
ctx = lib.Context(0)
dev = torch.device("cuda:0")
bases, off, lens = synth.ont_reads_device(20000, 20000 * 6000, 5_000_000, 0xC3, dev)
p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 200, A.SIG_U64, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
runs = [ctx.sketch(bases, off, p).cpu().numpy() for _ in range(4)]
for i in range(1, 4):
    d = np.nonzero((runs[0] != runs[i]).any(axis=1))[0]
    print("run", i, "rows differing:", d.size, d[:10], "lens", lens[d[:10]])
d = np.nonzero((runs[0] != runs[1]).any(axis=1))[0]
hb = bases.cpu().numpy(); ho = off.cpu().numpy().astype(np.uint64)
for r in d[:5]:
    b = hb[int(ho[r]):int(ho[r+1])]
    o2 = np.array([0, b.size], np.uint64)
    want = O.sketch(b.copy(), o2, p)[0]
    for i in range(4):
        w = np.nonzero(runs[i][r].view(np.uint64) != want)[0]
        print(" read", r, "len", lens[r], "run", i, "slots wrong vs oracle:", w[:8], w.size)
