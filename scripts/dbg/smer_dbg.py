import sys, numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, '.')
from kmerutils_amd import lib, _abi as A
from oracle import oracle as O
ctx = lib.Context(0)
rng = np.random.default_rng(3)
acgt = np.frombuffer(b"ACGT", np.uint8)
k = 31
for lens in ([200], [976], [977], [2000], [5000], [100, 100], [40, 40, 40], [1000] * 5):
    seqs = [bytes(acgt[rng.integers(0, 4, n)]) for n in lens]
    bases, off = O.concat(seqs)
    exp = sum(max(0, n - k + 1) for n in lens)
    for n_parts in (1, 8, 13):
        c = ctx.counter(A.KMER64BIT, k, 16, 1 << 16)
        rec, bounds, kmers = c.extract_superkmers(bases, off, n_parts)
        ctx.synchronize()
        r = rec.cpu().numpy().view(np.uint32).reshape(-1, 3)
        Ls = (r[:, 2] & 15) + 1
        print(lens if len(lens) < 4 else (lens[0], len(lens)), n_parts, "expected", exp, "census", int(kmers.sum()), "records", r.shape[0], "sumL", int(Ls.sum()))
        if n_parts == 1 and len(lens) == 1 and lens[0] <= 1000:
            print("   L:", Ls.tolist()[:80])
        c.close()
