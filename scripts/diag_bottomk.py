import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kmerutils_amd import _abi as A, lib, synth
from oracle import oracle as O
ctx = lib.Context(0)
kt, k, m, hs, fh = A.KMER16B32BIT, 16, 64, 2, 5
rng = np.random.default_rng(900 + m)
lens = [k - 1, k, k + 3, 100, 3000, 9000, 25000, 60000]
seqs = [bytes(synth.ACGT[rng.integers(0, 4, L)]) for L in lens]
for sub in ([3], [2, 3], [3, 4], [0, 1, 2, 3]):
    bases, off = O.concat([seqs[i] for i in sub])
    p = A.SketchParams(A.ALGO_BOTTOMK, kt, k, m, A.SIG_U64, hs, fh, 0, 0, 0, 0, 0)
    ws, wc = O.sketch(bases, off, p, want_counts=True)
    gs, gc = ctx.sketch(bases, off, p, want_counts=True)
    for r in range(len(sub)):
        bad = np.nonzero((gs[r] != ws[r]) | (gc[r] != wc[r]))[0]
        print("subset", sub, "read", sub[r], "bad idx", bad[:8])
        if bad.size:
            for i in bad[:4]:
                print("   idx", i, "got", hex(int(gs[r][i])), int(gc[r][i]), "want", hex(int(ws[r][i])), int(wc[r][i]), "prev want", hex(int(ws[r][i-1])))
print("done")
