"""debug: N ranks as threads, occurrences route, with / without a main-thread warm-up; checks what arrives"""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kmerutils_amd import _abi as A, dist as kdist, synth, lib
from oracle import oracle as O
import torch

world = int(sys.argv[1]) if len(sys.argv) > 1 else 6
warm = len(sys.argv) > 2 and sys.argv[2] == "warm"
os.environ["KMU_COUNT_ROUTE"] = sys.argv[3] if len(sys.argv) > 3 else "occurrences"
bases, off = synth.genome_reads(6000, np.full(6000, 150, np.int64), 40_000, 0xC4, sub=0.005, ins=0.0, dele=0.0)
if len(sys.argv) > 4 and sys.argv[4] == "ont":
    bases, off = synth.ont_reads(600, 500_000, 0xC3)
lens = np.diff(off.astype(np.int64))
shards = []
for r, (r0, r1) in enumerate(kdist.shard_reads_by_bases(lens, world)):
    shards.append((np.ascontiguousarray(bases[int(off[r0]):int(off[r1])]), (off[r0:r1 + 1] - off[r0]).astype(np.uint64), 60_000 if r % 2 else 200_000))
if warm:
    ctx = lib.Context(0)
    c = ctx.counter(A.KMER64BIT, 31, 16, 200_000)
    c.add_reads(torch.from_numpy(shards[0][0]).cuda(), torch.from_numpy(shards[0][1].astype(np.int64)).cuda())
    print("warm-up distinct", c.nb_distinct())
    c.close(); ctx.close()

class DbgTransport(kdist.ThreadTransport):
    def alltoallv(self, sp, sc, sd, rp, rc, rd, eb):
        W = self.g.world
        send = self._view(sp, max((sd[p] + sc[p]) * eb for p in range(W)))
        sums = [int(send[sd[p] * eb:(sd[p] + sc[p]) * eb].to(torch.int64).sum().item()) for p in range(W)]
        self.g.dbg[self.rank] = sums
        super().alltoallv(sp, sc, sd, rp, rc, rd, eb)
        recv = self._view(rp, max((rd[p] + rc[p]) * eb for p in range(W)))
        got = [int(recv[rd[p] * eb:(rd[p] + rc[p]) * eb].to(torch.int64).sum().item()) for p in range(W)]
        want = [self.g.dbg[p][self.rank] for p in range(W)]
        print("rank %d a2a eb=%d sc=%s rc=%s %s" % (self.rank, eb, sc, rc, "OK" if got == want else "MISMATCH got %s want %s" % (got, want)), flush=True)

group = kdist.ThreadGroup(world); group.dbg = [None] * world
res = [None] * world
def run(rank):
    try:
        torch.cuda.set_device(0)
        ctx = lib.Context(0)
        tt = DbgTransport(group, rank, torch.device("cuda", 0)); ctx._transport = tt
        ctx.comm_init_custom(rank, world, tt.alltoallv, tt.allgather)
        b, o, cap = shards[rank]
        c = ctx.counter(A.KMER64BIT, 31, 16, cap, distributed=True)
        print("rank %d table %s kmers %d" % (rank, c.table_info(), int(np.maximum(np.diff(o.astype(np.int64)) - 30, 0).sum())), flush=True)
        c.add_reads(torch.from_numpy(b).cuda(), torch.from_numpy(o.astype(np.int64)).cuda())
        c.finalize()
        res[rank] = c.nb_distinct()
    except BaseException as e:
        res[rank] = repr(e); group.barrier.abort(); raise
ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join(300) for t in ts]
print("result", res)
