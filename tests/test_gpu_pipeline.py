"""kmu_sketch_count: the reads once, both results.  Host form (chunked upload | sketch | download | count pipeline) and
device form, with plain and distributed counters, against the oracle; the chunking must not show in the results."""
import os

import numpy as np
import pytest

from kmerutils_amd import _abi as A
from kmerutils_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from kmerutils_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


def _check_counts(c, oracle, bases, off, k):
    g = oracle.Counter(A.KMER64BIT, k, 8, 1 << 20)
    g.add_reads(bases, off)
    wk, wc = g.dump(1)
    gk, gc = c.dump(1)
    assert np.array_equal(gk, wk) and np.array_equal(gc, wc)
    assert c.nb_occurrences() == int(np.maximum(np.diff(off.astype(np.int64)) - k + 1, 0).sum())


@pytest.mark.parametrize("pack", ["1", "0"])  # bases cross PCIe packed by the host's cores (the default of big calls) / as they are
@pytest.mark.parametrize("chunk_mb", ["1", "512"])
def test_sketch_count_host_and_device(ctx, oracle, monkeypatch, chunk_mb, pack):
    import torch
    monkeypatch.setenv("KMU_PIPE_CHUNK_MB", chunk_mb)
    monkeypatch.setenv("KMU_PIPE_PACK", pack)
    bases, off = synth.ont_reads(900, 600_000, 0xC3)  # ~5 Mbases: five chunks of 1 MB, reads of 200 .. 60 k bases
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 200, A.SIG_U64, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    want = oracle.sketch(bases, off, p)
    nk = int(np.maximum(np.diff(off.astype(np.int64)) - 30, 0).sum())
    # host buffers, pinned (torch) and pageable (numpy)
    for pinned in (True, False):
        c = ctx.counter(A.KMER64BIT, 31, 8, nk)
        if pinned:
            hb = torch.from_numpy(bases).pin_memory()
            ho = torch.from_numpy(off.astype(np.int64)).pin_memory()
            out = torch.zeros((len(off) - 1, 200), dtype=torch.int64).pin_memory()
            got = ctx.sketch_count(hb, ho, p, counter=c, out=out).numpy().view(np.uint64)
        else:
            got = ctx.sketch_count(bases, off, p, counter=c)
        assert np.array_equal(got, want)
        _check_counts(c, oracle, bases, off, 31)
        c.close()
    # a range of a larger read set (offsets[0] != 0), no counter
    got = ctx.sketch_count(bases, off[100:], p)
    assert np.array_equal(got, want[100:])
    # device buffers
    c = ctx.counter(A.KMER64BIT, 31, 8, nk)
    got = ctx.sketch_count(torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), p, counter=c)
    assert np.array_equal(got.cpu().numpy().view(np.uint64), want)
    _check_counts(c, oracle, bases, off, 31)
    c.close()
    # other sketchers ride the same pipeline
    ps = A.SketchParams(A.ALGO_SUPER, A.KMER64BIT, 21, 64, A.SIG_F64, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    assert np.array_equal(ctx.sketch_count(bases, off, ps).view(np.uint64), oracle.sketch(bases, off, ps).view(np.uint64))


@pytest.mark.parametrize("pack", ["1", "0"])
def test_sketch_count_errors_and_empty(ctx, oracle, monkeypatch, pack):
    from kmerutils_amd.lib import KmuError
    monkeypatch.setenv("KMU_PIPE_PACK", pack)  # (packed: the host's packer finds the byte; plain: the kernels do)
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 64, A.SIG_U64, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    bases = np.frombuffer(b"ACGT" * 20 + b"ACGTN" * 20, np.uint8).copy()
    off = np.array([0, 80, 180], np.uint64)
    with pytest.raises(KmuError) as ei:
        ctx.sketch_count(bases, off, p)
    assert ei.value.code == A.E_NON_ACGT
    # lower case is as good as upper case on either path (alphabet.rs:119-127)
    low = np.frombuffer(bytes(bases[:80]).lower() + bytes(bases[:80]), np.uint8).copy()
    o2 = np.array([0, 80, 160], np.uint64)
    rows = ctx.sketch_count(low, o2, p)
    assert np.array_equal(rows[0], rows[1]) and np.array_equal(rows, oracle.sketch(low, o2, p))
    pb = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 64, A.SIG_U64, 0, A.FHASH_CANON_INVHASH, 10, 0, 0, 0, 0)
    with pytest.raises(KmuError) as ei:
        ctx.sketch_count(bases[:80], off[:2], pb)
    assert ei.value.code == A.E_UNSUPPORTED
    # afterwards the context still works
    got = ctx.sketch_count(bases[:80], off[:2], p)
    assert np.array_equal(got, oracle.sketch(bases[:80], off[:2], p))


def test_sketch_count_distributed_world1(oracle, monkeypatch):
    """device form with a distributed counter: the all-to-all (RCCL, to self) is in flight while the reads are sketched"""
    import torch
    from kmerutils_amd import lib
    monkeypatch.setenv("NCCL_SOCKET_IFNAME", os.environ.get("NCCL_SOCKET_IFNAME", "lo"))
    ctx = lib.Context(0)
    ctx.comm_init(lib.Context.comm_get_id(), 0, 1)
    bases, off = synth.ont_reads(600, 500_000, 0xC3)
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 200, A.SIG_U64, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    want = oracle.sketch(bases, off, p)
    nk = int(np.maximum(np.diff(off.astype(np.int64)) - 30, 0).sum())
    for route in ("occurrences", "merge"):
        monkeypatch.setenv("KMU_COUNT_ROUTE", route)
        for dev in (True, False):
            c = ctx.counter(A.KMER64BIT, 31, 8, nk, distributed=True)
            if dev:
                got = ctx.sketch_count(torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), p, counter=c)
                got = got.cpu().numpy().view(np.uint64)
            else:
                got = ctx.sketch_count(bases, off, p, counter=c)
            c.finalize()
            assert np.array_equal(got, want)
            _check_counts(c, oracle, bases, off, 31)
            c.close()
    ctx.close()


@pytest.mark.parametrize("pack", ["1", "0"])
def test_sketch_count_host_amino_acids(ctx, oracle, monkeypatch, pack):
    """signatures-only kmu_sketch_count of protein sequences from HOST buffers: the packed upload packs 2-bit bases
    and would reject every residue outside ACGT -- amino-acid k-mers keep the plain upload, whatever KMU_PIPE_PACK says; the
    rows equal the oracle's (aautils path: src/aautils/setsketchert.rs:247-289)"""
    monkeypatch.setenv("KMU_PIPE_PACK", pack)
    monkeypatch.setenv("KMU_PIPE_CHUNK_MB", "1")
    res, off = synth.protein_seqs(6000, 0xA5)  # ~2 M residues: two chunks
    off = off.astype(np.uint64)
    for algo, sig, m in ((A.ALGO_SUPER, A.SIG_F64, 128), (A.ALGO_PROB3A, A.SIG_U64, 64)):
        p = A.SketchParams(algo, A.KMERAA64BIT, 12, m, sig, A.HASHER_NOHASH, A.FHASH_VALUE_MASKED, 0, 0, 0, 0, 0)
        want = oracle.sketch(res, off, p)
        got = np.asarray(ctx.sketch_count(res, off, p))
        assert np.array_equal(got.view(np.uint8), np.ascontiguousarray(want).view(np.uint8)), (algo, pack)


def test_sketch_count_host_chunked_level1(ctx, oracle, monkeypatch):
    """host form with the count's level-1 partition running chunk by chunk under the upload (the single-pass partition,
    forced on a batch the oracle can count: KMU_COUNT_SEG=2), nine 1 MB chunks; also with segments that overflow (the
    exact route redoes the whole batch at the end)"""
    import torch
    monkeypatch.setenv("KMU_PIPE_CHUNK_MB", "1")
    monkeypatch.setenv("KMU_PIPE_GROWTH", "1")  # (equal chunks: the packed upload's growing chunks would make fewer, larger arrivals)
    monkeypatch.setenv("KMU_COUNT_SEG", "2")
    monkeypatch.setenv("KMU_COUNT_SEG_ROUND_MIN", "1")  # (a launch per arrival of >= 1 wave step per unit: 256 KB here, 16 MB by default)
    dev = torch.device("cuda", 0)
    bases, off, lens = synth.ont_reads_device(1500, 9_000_000, 2_000_000, 0xC8, dev)
    hb, ho = bases[:int(off[-1])].cpu().numpy(), off.cpu().numpy().astype(np.uint64)
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 64, A.SIG_U64, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    want = oracle.sketch(hb, ho, p)
    o = oracle.Counter(A.KMER64BIT, 31, 8, 1 << 22)
    o.add_reads(hb, ho)
    wk, wc = o.dump(2)
    for pct in ("100", "60"):
        monkeypatch.setenv("KMU_COUNT_SEG_PCT", pct)
        c = ctx.counter(A.KMER64BIT, 31, 8, 16_000_000)
        ctx.profile_reset()
        ctx.profile_enable(True)
        got = ctx.sketch_count(hb, ho, p, counter=c)
        ctx.profile_enable(False)
        prof = ctx.profile_get()
        assert ("k_part_hist1" in prof) == (pct == "60")  # the histogram passes only run when the exact route takes over
        # (overflowing segments: the chunked attempt, then the exact levels -- no second attempt on the same k-mers)
        assert prof["k_arr_scatter"][0] == (1 if pct == "100" else 2)
        assert prof["k_part_scatter1"][0] >= 5  # the level-1 units took their slices of the stream in rounds, chunk by chunk
        assert np.array_equal(got, want)
        assert (c.nb_distinct(), c.nb_unique()) == (o.nb_distinct(), o.nb_unique())
        gk, gc = c.dump(2)
        assert np.array_equal(gk, wk) and np.array_equal(gc, wc)
        c.close()
