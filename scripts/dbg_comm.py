"""diagnostics: the OCCURRENCES route on one rank (RCCL to self) at a size where messages exceed 4 GiB, against the local build"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kmerutils_amd import _abi as A, lib, synth
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
n = int(os.environ.get("DBG_READS", "6250000"))
dev = torch.device("cuda", 0)
bases, off, lens = synth.ont_reads_device(n, n * 150, 100_000_000, 0xC4, dev, errors=(0.005, 0, 0), fixed_len=150)
nk = int(np.maximum(lens - 30, 0).sum())
ctx = lib.Context(0)
ctx.comm_init(lib.Context.comm_get_id(), 0, 1)
c0 = ctx.counter(A.KMER64BIT, 31, 8, nk)
c0.add_reads(bases, off)
want = (c0.nb_distinct(), c0.nb_unique(), c0.nb_occurrences())
c0.close()
for route in ("occurrences", "merge"):
    os.environ["KMU_COUNT_ROUTE"] = route
    for rep in range(2):
        c = ctx.counter(A.KMER64BIT, 31, 8, nk, distributed=True)
        t = time.time(); c.add_reads(bases, off); c.finalize(); ctx.synchronize(); dt = time.time() - t
        got = (c.nb_distinct(), c.nb_unique(), c.nb_occurrences())
        print(route, rep, "ok" if got == want else "MISMATCH", got, want, "%.1f ms" % (dt * 1e3), flush=True)
        c.close()
