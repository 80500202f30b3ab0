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
for lens in ([2000], [1000] * 5):
    seqs = [bytes(acgt[rng.integers(0, 4, n)]) for n in lens]
    bases, off = O.concat(seqs)
    h = O.kmer_hashes(bases, off, A.KMER64BIT, k, A.FHASH_CANON_VALUE)
    for n_parts in (1, 8):
        c = ctx.counter(A.KMER64BIT, k, 16, 1 << 16)
        rec, bounds, kmers = c.extract_superkmers(bases, off, n_parts)
        ctx.synchronize()
        r = rec.cpu().numpy().view(np.uint32).reshape(-1, 3)
        got, clean = O.superkmer_expand(r, k)
        gs = set(got.tolist())
        miss = []
        for i in range(len(off) - 1):
            for p in range(int(off[i]), int(off[i + 1]) - k + 1):
                if int(h[p]) not in gs:
                    miss.append(p)
        own = O.minimizer_owners(h[miss], k, n_parts) if miss else []
        print(lens[0], len(lens), n_parts, "missing positions:", miss, "owners", list(own))
        if n_parts == 8 and miss:
            p0 = miss[0]
            print("  owners around:", list(zip(range(p0 - 20, p0 + 20), O.minimizer_owners(h[p0 - 20:p0 + 20], k, 8).tolist())))
        c.close()
