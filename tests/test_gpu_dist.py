"""Two ranks sharing the one GPU of the test box (gloo transport): the real device code of the distributed counting
path -- owner grouping on the GPU, exchange, partitioned build of the owned table -- against the oracle."""
import os
import socket

import numpy as np
import pytest

from kmerutils_amd import _abi as A
from kmerutils_amd import dist as kdist

import host_exchange as hostx  # tests/host_exchange.py
from kmerutils_amd import synth

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    from kmerutils_amd import lib
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bases, off = synth.ont_reads(600, 500_000, 0xC3)  # ~3.5 Mbases, same set on both ranks
        lens = np.diff(off.astype(np.int64))
        r0, r1 = kdist.shard_reads_by_bases(lens, world)[rank]
        sb = torch.from_numpy(bases[int(off[r0]):int(off[r1])].copy()).cuda()
        so = torch.from_numpy((off[r0:r1 + 1] - off[r0]).astype(np.int64)).cuda()
        ctx = lib.Context(0)
        c = ctx.counter(A.KMER64BIT, 31, 8, int(off[-1]))
        # the shard is sketched while the exchange is in flight (what bench.py does at N > 1), with CUs held back for it
        os.environ["KMU_PMH_RESERVE_CUS"] = "16"
        p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 64, A.SIG_U64, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
        box = {}
        nrecv = hostx.count_reads_exchange(c, sb, so, overlap=lambda: box.update(sig=ctx.sketch(sb, so, p)))
        g = O.Counter(A.KMER64BIT, 31, 8, 1 << 20)
        g.add_reads(bases, off)
        gk, gc = g.dump(1)
        L = O.lib()
        own = np.array([L.kmo_int64_hash(int(x)) % world for x in gk], dtype=np.int64) == rank
        kk, cc = c.export_part(0, 1)
        order = np.argsort(kk)
        ok = nrecv > 0 and np.array_equal(kk[order], gk[own]) and np.array_equal(np.minimum(cc[order], 255), gc[own])
        # sketch shards: rows of this rank == rows r0..r1 of the single-process result
        mine = box["sig"].cpu().numpy().view(np.uint64)
        want = O.sketch(bases, off, p)[r0:r1]
        ok = ok and np.array_equal(mine, want)
        # ONE signature for the union of both ranks' reads (sketch_compressedkmer_seqs across ranks): the weighted sketch
        # goes through the owner exchange, the unweighted ones through an all-gather of per-rank minima
        for algo, sig in ((A.ALGO_PROB3A, A.SIG_U64), (A.ALGO_SUPER, A.SIG_F64), (A.ALGO_OPTDENS, A.SIG_F64), (A.ALGO_SUPER2, A.SIG_U64)):
            pa = A.SketchParams(algo, A.KMER64BIT, 31, 96, sig, 0, A.FHASH_CANON_INVHASH, 0, A.MODE_ALL_SEQS, 0, 0, 0)
            got = kdist.sketch_seqs_distributed(ctx, sb, so, pa)
            want_all = O.sketch(bases, off, pa)[0]
            ok = ok and np.array_equal(got.cpu().numpy().view(np.uint8), np.ascontiguousarray(want_all).view(np.uint8))
        ret[rank] = bool(ok)
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_exchange():
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
        p.join(600)
        assert p.exitcode == 0
    assert ret.get(0) is True and ret.get(1) is True
