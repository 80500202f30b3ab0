"""The N > 1 path on CPU: world_size-2 gloo process group, read sharding and the owner-partition exchange of
kmerutils_amd.dist, with a test-only counter double backed by the oracle standing in for the GPU table."""
import os
import socket

import numpy as np
import pytest

from kmerutils_amd import _abi as A
from kmerutils_amd import dist as kdist

import host_exchange as hostx  # tests/host_exchange.py
from kmerutils_amd import synth


def test_shard_reads_by_bases():
    rng = np.random.default_rng(1)
    lens = synth.ont_lengths(5000, rng)
    for w in (1, 2, 3, 8):
        sh = kdist.shard_reads_by_bases(lens, w)
        assert sh[0][0] == 0 and sh[-1][1] == len(lens)
        assert all(sh[i][1] == sh[i + 1][0] for i in range(w - 1))
        per = np.array([lens[a:b].sum() for a, b in sh])
        assert per.max() - per.min() <= 2 * lens.max()
    assert kdist.shard_reads_by_bases([], 2) == [(0, 0), (0, 0)]
    assert kdist.shard_reads_by_bases([5], 4)[-1][1] == 1


class OracleCounterDouble:
    """test-only stand-in with the lib.Counter surface used by tests/host_exchange.merge_counters"""

    def __init__(self, O, kmer_type, k):
        self.O, self.kt, self.k = O, kmer_type, k
        self.c = O.Counter(kmer_type, k, 16, 1 << 12)

    def add_reads(self, bases, offsets):
        self.c.add_reads(bases, offsets)

    def export_part(self, part, n_parts, device=None):
        k, c = self.c.dump(1)
        L = self.O.lib()
        own = np.array([L.kmo_int64_hash(int(x)) % n_parts for x in k], dtype=np.int64) == part
        return k[own].copy(), c[own].copy()

    def reset(self):
        self.c = self.O.Counter(self.kt, self.k, 16, 1 << 12)

    def merge_entries(self, k, c):
        self.c.add_kmers(np.repeat(k, c.astype(np.int64)))

    def add_kmers(self, k):
        self.c.add_kmers(np.ascontiguousarray(k, np.uint64))

    def extract_by_owner(self, bases, offsets, n_parts):
        import torch
        canon = self.O.kmer_hashes(bases, offsets, self.kt, self.k, A.FHASH_CANON_VALUE)
        L = np.diff(offsets.astype(np.int64))
        keep = np.concatenate([np.arange(int(offsets[i]), int(offsets[i]) + max(0, int(L[i]) - self.k + 1))
                               for i in range(len(L))]) if len(L) else np.zeros(0, np.int64)
        km = canon[keep]
        lib = self.O.lib()
        own = np.array([lib.kmo_int64_hash(int(x)) % n_parts for x in km], dtype=np.int64)
        order = np.argsort(own, kind="stable")
        bounds = np.zeros(n_parts + 1, np.uint64)
        bounds[1:] = np.cumsum(np.bincount(own, minlength=n_parts))
        return torch.from_numpy(km[order].view(np.int64).copy()), bounds


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bases, off = synth.illumina_reads(400, 3000, 0xC4)  # every rank builds the same read set ...
        lens = np.diff(off.astype(np.int64))
        r0, r1 = kdist.shard_reads_by_bases(lens, world)[rank]  # ... and counts only its contiguous shard
        sb = bases[int(off[r0]):int(off[r1])].copy()
        so = (off[r0:r1 + 1] - off[r0]).astype(np.uint64)
        cd = OracleCounterDouble(O, A.KMER64BIT, 21)
        cd.add_reads(sb, so)
        got = hostx.merge_counters(cd, device=None, chunk_entries=257)  # several exchange rounds
        # reference result: the global counter restricted to the keys this rank owns
        g = O.Counter(A.KMER64BIT, 21, 16, 1 << 12)
        g.add_reads(bases, off)
        gk, gc = g.dump(1)
        L = O.lib()
        own = np.array([L.kmo_int64_hash(int(x)) % world for x in gk], dtype=np.int64) == rank
        mk, mc = cd.c.dump(1)
        ok = np.array_equal(mk, gk[own]) and np.array_equal(mc, gc[own]) and got > 0
        # signature slabs are concatenated in rank (= read) order
        import torch
        rows = torch.arange((r1 - r0) * 4, dtype=torch.int64).reshape(r1 - r0, 4) + 1000 * rank
        allrows = kdist.gather_rows(rows)
        ok = ok and allrows.shape[0] == len(lens) and int(allrows[0, 0]) == 0
        # the throughput path: k-mers grouped by owner, one all-to-all, owner builds its table
        cd2 = OracleCounterDouble(O, A.KMER64BIT, 21)
        nrecv = hostx.count_reads_exchange(cd2, sb, so)
        k2, c2 = cd2.c.dump(1)
        ok = ok and nrecv > 0 and np.array_equal(k2, gk[own]) and np.array_equal(c2, gc[own])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_merge_counters_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(0) is True and ret.get(1) is True


# ---- the transport handed to kmu_comm_init_custom (dist.TorchTransport) on host buffers, world 2, gloo ------------------------
def _transport_worker(rank, world, port, ret):
    import ctypes as C
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tt = kdist.TorchTransport(None, None)
        ok = tt.allgather(bytes([rank, 7, rank + 1])) == bytes([0, 7, 1, 1, 7, 2])
        # rank r sends (r + 1) * (p + 1) elements of 8 bytes to peer p, values tagged with (sender, receiver, index)
        for eb, dt in ((8, np.uint64), (4, np.uint32)):
            sc = [(rank + 1) * (p + 1) for p in range(world)]
            sd = [int(x) for x in np.concatenate([[0], np.cumsum(sc)[:-1]])]
            rc = [(p + 1) * (rank + 1) for p in range(world)]
            rd = [int(x) for x in np.concatenate([[0], np.cumsum(rc)[:-1]])]
            send = np.concatenate([np.arange(sc[p], dtype=dt) + 1000 * rank + 100 * p for p in range(world)])
            recv = np.zeros(sum(rc), dt)
            tt.alltoallv(send.ctypes.data, sc, sd, recv.ctypes.data, rc, rd, eb)
            want = np.concatenate([np.arange(rc[p], dtype=dt) + 1000 * p + 100 * rank for p in range(world)])
            ok = ok and np.array_equal(recv, want)
        # an empty exchange is legal (a rank without k-mers)
        tt.alltoallv(0, [0] * world, [0] * world, 0, [0] * world, [0] * world, 8)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_torch_transport_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_transport_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(0) is True and ret.get(1) is True


def test_kmer_owner_matches_oracle(oracle):
    """kmu_kmer_owner = DispatchableT::dispatch (kmercount.rs:382-420): int32_hash / int64_hash of the value, mod n"""
    from kmerutils_amd import lib
    L = oracle.lib()
    rng = np.random.default_rng(5)
    k64 = rng.integers(0, 1 << 62, 500, dtype=np.uint64)
    for n in (1, 2, 3, 8, 13):
        assert list(lib.kmer_owner(A.KMER64BIT, k64, n)) == [L.kmo_int64_hash(int(x)) % n for x in k64]
        k32 = (k64 & np.uint64(0xFFFFFFFF))
        assert list(lib.kmer_owner(A.KMER16B32BIT, k32, n)) == [L.kmo_int32_hash(int(x)) % n for x in k32]
