"""GPU parity: the HIP path (through the C-ABI of libkmu.so) against the CPU oracle on the same seeded inputs.
Bit-exact for every integer / byte / index output; the f32/f64 SuperMinHash values are compared bit for bit as
well (same IEEE operations in the same order, -ffp-contract=off)."""
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


def ragged_dna(seed, lens):
    rng = np.random.default_rng(seed)
    seqs = [bytes(synth.ACGT[rng.integers(0, 4, L)]) for L in lens]
    return seqs


RAGGED = [1, 2, 7, 15, 16, 17, 30, 31, 32, 33, 47, 48, 49, 63, 64, 65, 100, 150, 151, 1000, 1023, 1024, 1025, 1039,
          1040, 1041, 2047, 2048, 2079, 5000, 12345]


@pytest.mark.parametrize("kmer_type,k", [(A.KMER32BIT, 3), (A.KMER32BIT, 8), (A.KMER32BIT, 14), (A.KMER16B32BIT, 16),
                                         (A.KMER64BIT, 17), (A.KMER64BIT, 21), (A.KMER64BIT, 31)])
def test_kmer_hashes_all_fhash(ctx, oracle, kmer_type, k):
    seqs = ragged_dna(100 + k, RAGGED)
    bases, off = oracle.concat(seqs)
    for fh in range(8):
        want = oracle.kmer_hashes(bases, off, kmer_type, k, fh)
        got = ctx.kmer_hashes(bases, off, kmer_type, k, fh)
        assert np.array_equal(got, want), "fhash %d" % fh


def test_kmer_hashes_packed_and_lowercase(ctx, oracle):
    seqs = ragged_dna(5, RAGGED)
    bases, off = oracle.concat(seqs)
    packed, poff = ctx.pack2b(bases, off)
    # packing itself: Sequence::new(raw, 2)
    for i, s in enumerate(seqs):
        assert bytes(packed[int(poff[i]):int(poff[i + 1])]) == bytes(oracle.pack2b(s))
    for kmer_type, k in ((A.KMER32BIT, 11), (A.KMER16B32BIT, 16), (A.KMER64BIT, 31)):
        for fh in (A.FHASH_IDENTITY_RAW, A.FHASH_CANON_INVHASH, A.FHASH_CANON_NTHASH):
            want = oracle.kmer_hashes(bases, off, kmer_type, k, fh)
            got = ctx.kmer_hashes(packed, off, kmer_type, k, fh, A.INPUT_PACKED2, poff)
            assert np.array_equal(got, want)
    lower = np.frombuffer(bytes(bases).lower(), np.uint8).copy()
    assert np.array_equal(ctx.kmer_hashes(lower, off, A.KMER64BIT, 21, A.FHASH_CANON_INVHASH),
                          oracle.kmer_hashes(bases, off, A.KMER64BIT, 21, A.FHASH_CANON_INVHASH))


def test_non_acgt_and_errors(ctx, oracle):
    from kmerutils_amd.lib import KmuError
    seqs = [b"ACGTACGTNACGT", b"ACGT" * 10, b"acgtxx", b"A" * 100 + b"N"]
    bases, off = oracle.concat(seqs)
    assert ctx.count_non_acgt(bases, off).tolist() == [1, 0, 2, 1]
    with pytest.raises(KmuError) as e:
        ctx.kmer_hashes(bases, off, A.KMER32BIT, 5, A.FHASH_IDENTITY_RAW)
    assert e.value.code == A.E_NON_ACGT
    with pytest.raises(KmuError) as e:
        ctx.pack2b(bases, off)
    assert e.value.code == A.E_NON_ACGT
    good, goff = oracle.concat([b"ACGT" * 10])
    for kt, k in ((A.KMER32BIT, 15), (A.KMER16B32BIT, 15), (A.KMER64BIT, 32), (A.KMERAA64BIT, 13)):
        with pytest.raises(KmuError) as e:
            ctx.kmer_hashes(good, goff, kt, k, A.FHASH_IDENTITY_RAW)
        assert e.value.code == A.E_BAD_K
    # empty sequence: the reference panics (nbkmerguess.rs:8)
    eb, eo = oracle.concat([b"ACGTACGTACGT", b""])
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER32BIT, 5, 16, A.SIG_U32, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    with pytest.raises(KmuError) as e:
        ctx.sketch(eb, eo, p)
    assert e.value.code == A.E_EMPTY_SEQ
    with pytest.raises(oracle.OracleError):
        oracle.sketch(eb, eo, p)


def test_aa_kmers(ctx, oracle):
    res, off = synth.protein_seqs(200, 11, median=120)
    for kt, k in ((A.KMERAA32BIT, 4), (A.KMERAA32BIT, 6), (A.KMERAA64BIT, 8), (A.KMERAA64BIT, 12)):
        for fh in (A.FHASH_IDENTITY_RAW, A.FHASH_VALUE_MASKED, A.FHASH_INVHASH_RAW):
            assert np.array_equal(ctx.kmer_hashes(res, off, kt, k, fh), oracle.kmer_hashes(res, off, kt, k, fh))
    from kmerutils_amd.lib import KmuError
    bad, boff = oracle.concat([b"MTEQBELIK"])
    with pytest.raises(KmuError) as e:
        ctx.kmer_hashes(bad, boff, A.KMERAA32BIT, 3, A.FHASH_IDENTITY_RAW)
    assert e.value.code == A.E_BAD_ALPHABET


PMH_CASES = [
    # (kmer_type, k, m, fhash, lens)
    (A.KMER32BIT, 8, 200, A.FHASH_CANON_INVHASH, [5, 8, 9, 200, 1000, 5900, 20000]),
    (A.KMER32BIT, 5, 64, A.FHASH_IDENTITY_RAW, [3, 4, 5, 6, 50, 3000]),
    (A.KMER16B32BIT, 16, 100, A.FHASH_CANON_INVHASH, [15, 16, 17, 1000, 9000]),
    (A.KMER64BIT, 31, 200, A.FHASH_CANON_INVHASH, [30, 31, 32, 150, 6000, 9000, 30000, 70000]),
    (A.KMER64BIT, 21, 2, A.FHASH_CANON_NTHASH, [21, 500]),
    (A.KMER64BIT, 31, 257, A.FHASH_CANON_RAW, [4000, 100]),
]


@pytest.mark.parametrize("kmer_type,k,m,fhash,lens", PMH_CASES)
def test_probminhash3a_parity(ctx, oracle, kmer_type, k, m, fhash, lens):
    seqs = ragged_dna(1000 + k + m, lens)
    # a low-complexity read: heavy weights
    seqs.append(b"ACGT" * 700 + b"A" * 900 + bytes(synth.ACGT[np.random.default_rng(3).integers(0, 4, 400)]))
    # repetitive reads whose k-mer occurrences exceed the dense LDS arrays: redone in rounds with a carry list
    seqs.append(b"A" * 40000 + b"ACGT" * 6000 + bytes(synth.ACGT[np.random.default_rng(4).integers(0, 4, 9000)]) + b"T" * 30000)
    seqs.append((bytes(synth.ACGT[np.random.default_rng(6).integers(0, 4, 7000)]) + b"GATTACA" * 300) * 6)
    bases, off = oracle.concat(seqs)
    sig_t = A.SIG_U32 if A.kmer_val_bytes(kmer_type) == 4 else A.SIG_U64
    for flags in (0, A.FLAG_RAND08):
        p = A.SketchParams(A.ALGO_PROB3A, kmer_type, k, m, sig_t, A.HASHER_NOHASH, fhash, 0, 0, 0, 0, flags)
        want = oracle.sketch(bases, off, p)
        got = ctx.sketch(bases, off, p)
        assert got.dtype == want.dtype
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, "rows differ: %s" % bad[:5]
    # reads shorter than k: all-zero signature (empty map -> [initobj; m])
    short = [i for i, L in enumerate(lens) if L < k]
    for i in short:
        assert not got[i].any()


def test_probminhash3a_ont_batch_and_device_memory(ctx, oracle):
    import torch
    bases, off = synth.ont_reads(300, 2_000_000, 0xC3)
    for kt, k, sig in ((A.KMER32BIT, 8, A.SIG_U32), (A.KMER64BIT, 31, A.SIG_U64)):
        p = A.SketchParams(A.ALGO_PROB3A, kt, k, 200, sig, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
        want = oracle.sketch(bases, off, p)
        got = ctx.sketch(bases, off, p)
        assert np.array_equal(got, want)
        # same through device-resident buffers (torch tensors), twice: determinism
        tb = torch.from_numpy(bases).cuda()
        to = torch.from_numpy(off.astype(np.int64)).cuda()
        g1 = ctx.sketch(tb, to, p).cpu().numpy()
        g2 = ctx.sketch(tb, to, p).cpu().numpy()
        assert np.array_equal(g1.view(want.dtype), want) and np.array_equal(g1, g2)


def test_block_sketch_parity(ctx, oracle):
    seqs = ragged_dna(77, [10, 999, 1000, 1001, 2500, 7, 8, 30000])
    bases, off = oracle.concat(seqs)
    for B in (1000, 257, 4096):
        p = A.SketchParams(A.ALGO_PROB3A, A.KMER32BIT, 8, 50, A.SIG_U32, 0, A.FHASH_CANON_INVHASH, B, 0, 0, 0, 0)
        want = oracle.sketch(bases, off, p)
        got = ctx.sketch(bases, off, p)
        assert got.shape == want.shape and np.array_equal(got, want)
        # empty trailing blocks give all-zero rows (seqblocksketch.rs:137-145)
        bro = oracle.block_layout(off, B)
        assert np.array_equal(ctx.block_layout(off, B), bro)


SUPER_CASES = [
    (A.ALGO_SUPER, A.KMER16B32BIT, 16, 64, A.SIG_F64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH),
    (A.ALGO_SUPER, A.KMER16B32BIT, 16, 64, A.SIG_F32, A.HASHER_FNV1A, A.FHASH_CANON_INVHASH),
    (A.ALGO_SUPER, A.KMER32BIT, 10, 100, A.SIG_F64, A.HASHER_FNV1A, A.FHASH_IDENTITY_RAW),
    (A.ALGO_SUPER, A.KMER64BIT, 31, 128, A.SIG_F64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH),
    (A.ALGO_SUPER, A.KMER64BIT, 21, 300, A.SIG_F32, A.HASHER_NOHASH, A.FHASH_VALUE_MASKED),
    (A.ALGO_SUPER2, A.KMER64BIT, 21, 128, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH),
    (A.ALGO_SUPER2, A.KMER16B32BIT, 16, 50, A.SIG_U32, A.HASHER_FNV1A, A.FHASH_CANON_INVHASH),
]


@pytest.mark.parametrize("algo,kmer_type,k,m,sig,hasher,fhash", SUPER_CASES)
def test_superminhash_parity(ctx, oracle, algo, kmer_type, k, m, sig, hasher, fhash):
    seqs = ragged_dna(500 + m, [k - 1, k, k + 1, 100, 1000, 1000, 3000, 10000])
    bases, off = oracle.concat(seqs)
    for flags in (0, A.FLAG_RAND08):
        p = A.SketchParams(algo, kmer_type, k, m, sig, hasher, fhash, 0, 0, 0, 0, flags)
        want = oracle.sketch(bases, off, p)
        got = ctx.sketch(bases, off, p)
        assert got.dtype == want.dtype
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


@pytest.mark.parametrize("kmer_type,k,m,hasher,fhash", [
    (A.KMER16B32BIT, 16, 50, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH),     # sketch_seqrange_minhash, seqminhash.rs:65-119
    (A.KMER32BIT, 12, 200, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH),
    (A.KMER16B32BIT, 16, 64, A.HASHER_INT64HASH, A.FHASH_CANON_VALUE),   # MinInvHashCountKmer, minhash.rs:219-265
    (A.KMER64BIT, 31, 1000, A.HASHER_INT64HASH, A.FHASH_CANON_VALUE),
    (A.KMER64BIT, 21, 300, A.HASHER_FNV1A, A.FHASH_IDENTITY_RAW),
    (A.KMER32BIT, 6, 5000, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH),       # fewer distinct k-mers than the sketch size
])
def test_bottomk_parity(ctx, oracle, kmer_type, k, m, hasher, fhash):
    seqs = ragged_dna(900 + m, [k - 1, k, k + 3, 100, 3000, 9000, 25000, 60000])
    seqs.append(b"ACGT" * 2000 + b"A" * 70000 + bytes(synth.ACGT[np.random.default_rng(5).integers(0, 4, 3000)]))
    bases, off = oracle.concat(seqs)
    p = A.SketchParams(A.ALGO_BOTTOMK, kmer_type, k, m, A.SIG_U64, hasher, fhash, 0, 0, 0, 0, 0)
    wsig, wcnt = oracle.sketch(bases, off, p, want_counts=True)
    gsig, gcnt = ctx.sketch(bases, off, p, want_counts=True)
    assert np.array_equal(gsig, wsig)
    assert np.array_equal(gcnt, wcnt)
    assert np.array_equal(ctx.sketch(bases, off, p), wsig)  # counts not requested


@pytest.mark.parametrize("algo,kmer_type,k,m,sig,hasher", [
    (A.ALGO_PROB3A, A.KMER32BIT, 8, 200, A.SIG_U32, A.HASHER_NOHASH),
    (A.ALGO_PROB3A, A.KMER64BIT, 31, 128, A.SIG_U64, A.HASHER_NOHASH),
    (A.ALGO_SUPER, A.KMER16B32BIT, 16, 64, A.SIG_F64, A.HASHER_NOHASH),
    (A.ALGO_SUPER, A.KMER64BIT, 21, 100, A.SIG_F32, A.HASHER_FNV1A),
    (A.ALGO_SUPER2, A.KMER64BIT, 21, 128, A.SIG_U64, A.HASHER_NOHASH),
])
def test_all_seqs_mode_parity(ctx, oracle, algo, kmer_type, k, m, sig, hasher):
    """sketch_compressedkmer_seqs: ONE signature for a whole list of sequences (setsketchert.rs:160-202, :299-335);
    ~600 k k-mers => many LDS-sized leaves / chunks merged on the device"""
    bases, off = synth.ont_reads(120, 400_000, 0xA11)
    extra, eoff = oracle.concat([b"A" * 3000, b"ACGT" * 500, b"AC"])
    bases = np.concatenate([bases, extra])
    off = np.concatenate([off, eoff[1:] + off[-1]])
    p = A.SketchParams(algo, kmer_type, k, m, sig, hasher, A.FHASH_CANON_INVHASH, 0, A.MODE_ALL_SEQS, 0, 0, 0)
    want = oracle.sketch(bases, off, p)
    got = ctx.sketch(bases, off, p)
    assert got.shape == want.shape == (1, m)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    small, soff = oracle.concat([b"ACGTTGCA" * 40, b"GATTACAGATTACA" * 9])
    assert np.array_equal(ctx.sketch(small, soff, p).view(np.uint8), oracle.sketch(small, soff, p).view(np.uint8))


def test_sketch_hashed_parity(ctx, oracle):
    """kmu_sketch_hashed: the host evaluates an arbitrary fhash closure, the device does multiset + sketch"""
    rng = np.random.default_rng(21)
    lens = [0, 1, 50, 4000, 30000, 9000]
    off = np.zeros(len(lens) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    for dt, kt in ((np.uint32, A.KMER16B32BIT), (np.uint64, A.KMER64BIT)):
        vals = rng.integers(0, 5000 if dt == np.uint32 else 2**62, int(off[-1]), dtype=np.uint64).astype(dt)
        vals[100:3000] = vals[100]  # heavy repeats
        sigt = A.SIG_U32 if dt == np.uint32 else A.SIG_U64
        cases = [(A.ALGO_PROB3A, sigt, A.HASHER_NOHASH, A.MODE_PER_SEQ), (A.ALGO_SUPER, A.SIG_F64, A.HASHER_FNV1A, A.MODE_PER_SEQ),
                 (A.ALGO_BOTTOMK, A.SIG_U64, A.HASHER_INT64HASH, A.MODE_PER_SEQ), (A.ALGO_PROB3A, sigt, A.HASHER_NOHASH, A.MODE_ALL_SEQS),
                 (A.ALGO_SUPER, A.SIG_F32, A.HASHER_NOHASH, A.MODE_ALL_SEQS)]
        for algo, sig, hasher, mode in cases:
            p = A.SketchParams(algo, kt, 16 if dt == np.uint32 else 31, 100, sig, hasher, 0, 0, mode, 0, 0, 0)
            want = oracle.sketch_hashed(vals, off, p)
            got = ctx.sketch_hashed(vals, off, p)
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (algo, mode, dt)


def test_count_add_kmers_partitioned(ctx, oracle):
    """explicit canonical k-mers in one big batch go through the radix-partitioned build as well"""
    bases, off = synth.illumina_reads(20000, 200_000, 0xC4)
    canon = oracle.kmer_hashes(bases, off, A.KMER64BIT, 31, A.FHASH_CANON_VALUE)
    L = np.diff(off.astype(np.int64))
    keep = np.concatenate([np.arange(int(off[i]), int(off[i]) + max(0, int(L[i]) - 30)) for i in range(len(L))])
    kmers = np.ascontiguousarray(canon[keep])
    oc = oracle.Counter(A.KMER64BIT, 31, 8, 1 << 20)
    oc.add_kmers(kmers)
    gc = ctx.counter(A.KMER64BIT, 31, 8, kmers.size)  # batch >= table / 4 -> partitioned
    gc.add_kmers(kmers)
    assert gc.nb_distinct() == oc.nb_distinct() and gc.nb_unique() == oc.nb_unique()
    gk, gcn = gc.dump(2)
    wk, wc = oc.dump(2)
    assert np.array_equal(gk, wk) and np.array_equal(gcn, wc)


def test_extract_by_owner_single_gpu(ctx, oracle):
    """the multi-GPU exchange front end on one GPU: k-mers grouped by owner rank, every group complete and pure"""
    bases, off = synth.illumina_reads(4000, 100_000, 0xC4)
    c = ctx.counter(A.KMER64BIT, 31, 8, 1 << 20)
    world = 5
    kmers, bounds = c.extract_by_owner(bases, off, world)
    km = kmers.cpu().numpy().view(np.uint64)
    L = oracle.lib()
    canon = oracle.kmer_hashes(bases, off, A.KMER64BIT, 31, A.FHASH_CANON_VALUE)
    lens = np.diff(off.astype(np.int64))
    keep = np.concatenate([np.arange(int(off[i]), int(off[i]) + max(0, int(lens[i]) - 30)) for i in range(len(lens))])
    want = canon[keep]
    assert km.size == want.size == int(bounds[-1])
    for p in range(world):
        grp = km[int(bounds[p]):int(bounds[p + 1])]
        own = np.array([L.kmo_int64_hash(int(x)) % world for x in grp[:500]])
        assert (own == p).all()
    assert np.array_equal(np.sort(km), np.sort(want))
    # each owner building from its group reproduces the global counts restricted to its keys
    oc = oracle.Counter(A.KMER64BIT, 31, 8, 1 << 20)
    oc.add_kmers(want)
    gk, gc = oc.dump(1)
    for p in (0, world - 1):
        cp = ctx.counter(A.KMER64BIT, 31, 8, 1 << 20)
        cp.add_kmers(np.ascontiguousarray(km[int(bounds[p]):int(bounds[p + 1])]))
        kk, cc = cp.dump(1)
        own = np.array([L.kmo_int64_hash(int(x)) % world for x in gk]) == p
        assert np.array_equal(kk, gk[own]) and np.array_equal(cc, gc[own])


def test_superminhash_aa(ctx, oracle):
    res, off = synth.protein_seqs(300, 0xC5, median=250)
    for kt, k, m in ((A.KMERAA64BIT, 12, 128), (A.KMERAA32BIT, 5, 400)):
        p = A.SketchParams(A.ALGO_SUPER, kt, k, m, A.SIG_F64, A.HASHER_NOHASH, A.FHASH_VALUE_MASKED, 0, 0, 0, 0, 0)
        assert np.array_equal(ctx.sketch(res, off, p), oracle.sketch(res, off, p))
        p2 = A.SketchParams(A.ALGO_PROB3A, kt, k, 64, A.SIG_U64 if kt == A.KMERAA64BIT else A.SIG_U32, 0,
                            A.FHASH_VALUE_MASKED, 0, 0, 0, 0, 0)
        assert np.array_equal(ctx.sketch(res, off, p2), oracle.sketch(res, off, p2))


@pytest.mark.parametrize("kmer_type,k", [(A.KMER64BIT, 21), (A.KMER64BIT, 31), (A.KMER16B32BIT, 16), (A.KMER32BIT, 12)])
def test_count_parity(ctx, oracle, kmer_type, k):
    _count_parity(ctx, oracle, kmer_type, k)


def _count_parity(ctx, oracle, kmer_type, k):
    bases, off = synth.illumina_reads(3000, 20000, 0xC2)  # ~22x coverage: multiplicities well above 2
    extra, eoff = oracle.concat([b"A" * 400, b"ACGT" * 5, b"AC", b"T" * 399])
    bases = np.concatenate([bases, extra])
    off = np.concatenate([off, eoff[1:] + off[-1]])
    oc = oracle.Counter(kmer_type, k, 8, 1 << 16)
    oc.add_reads(bases, off)
    gc = ctx.counter(kmer_type, k, 8, 1 << 16)
    gc.add_reads(bases, off)
    assert gc.nb_distinct() == oc.nb_distinct()
    assert gc.nb_unique() == oc.nb_unique()
    wk, wc = oc.dump(2)
    gk, gcn = gc.dump(2)
    assert np.array_equal(gk, wk) and np.array_equal(gcn, wc)
    assert wc.max() == 255  # poly-A saturates the 8-bit counter
    canon = oracle.kmer_hashes(bases, off, kmer_type, k, A.FHASH_CANON_VALUE)
    q = np.concatenate([canon[:5000], np.arange(1000, dtype=np.uint64) * 7919 + 3])
    assert np.array_equal(gc.query(q), oc.query(q))
    # a second batch accumulates; packed input goes through the per-read kernel
    packed, poff = ctx.pack2b(bases, off)
    gc.add_reads(packed, off, A.INPUT_PACKED2, poff)
    oc.add_reads(bases, off)
    assert np.array_equal(gc.query(q), oc.query(q))
    # key-partition export / merge / retain (multi-GPU merge building blocks)
    g2 = ctx.counter(kmer_type, k, 16, 1 << 16)
    tot = 0
    for part in range(3):
        kk, cc = gc.export_part(part, 3)
        tot += kk.size
        g2.merge_entries(kk, cc)
    assert tot == gc.nb_distinct() == g2.nb_distinct()
    o16 = oracle.Counter(kmer_type, k, 16, 1 << 16)
    o16.add_reads(bases, off)
    o16.add_reads(bases, off)
    # (what an 8-bit counter exports is exact up to the ceiling of its table's count field: 32 bits wide in the 12-byte slot
    #  format, 2^w - 1024 >= 255 in the 8-byte one -- the reference's own 8-bit counters stop at 255, kmercount.rs:1615)
    ti = gc.table_info()
    ceil = ti["count_ceiling"] if ti["bytes_per_slot"] == 8 else 1 << 32
    assert ti["count_ceiling"] == ((1 << ti["count_field_bits"]) - 1024 if ti["bytes_per_slot"] == 8 else (1 << 32) - 1)
    assert np.array_equal(g2.query(q), np.minimum(o16.query(q), min(ceil, 65535)))
    gc.retain_part(1, 3)
    k1, c1 = g2.export_part(1, 3)
    assert gc.nb_distinct() == k1.size


def test_count_partitioned_two_level_vs_direct_and_oracle(ctx, oracle):
    """the radix-partitioned build (two levels: table >= 2^24 slots) against the direct-insert path and the oracle,
    including a second batch merged into a non-empty table"""
    import os
    bases, off = synth.ont_reads(2500, 3_000_000, 0xC4)  # ~15 Mbases
    extra, eoff = oracle.concat([b"A" * 5000, b"ACGT" * 100, b"AC", b"T" * 3000])
    bases = np.concatenate([bases, extra])
    off = np.concatenate([off, eoff[1:] + off[-1]])
    k = 31
    oc = oracle.Counter(A.KMER64BIT, k, 8, 1 << 20)
    oc.add_reads(bases, off)
    res = {}
    for path in ("partitioned", "direct"):
        os.environ["KMU_COUNT_PATH"] = path
        try:
            gc = ctx.counter(A.KMER64BIT, k, 8, 12_000_000)  # 2^25 slots -> 13 region bits -> two levels
            gc.add_reads(bases, off)
            assert gc.nb_distinct() == oc.nb_distinct() and gc.nb_unique() == oc.nb_unique()
            res[path] = gc.dump(2)
            if path == "partitioned":
                gc.add_reads(bases, off)  # non-empty table: regions are loaded, updated, written back
                kk, cc = gc.export_part(0, 1)
                k1, c1 = oc.dump(1)
                order = np.argsort(kk)
                assert np.array_equal(kk[order], k1)
                oc2 = oracle.Counter(A.KMER64BIT, k, 16, 1 << 20)
                oc2.add_reads(bases, off)
                oc2.add_reads(bases, off)
                canon = oracle.kmer_hashes(bases, off, A.KMER64BIT, k, A.FHASH_CANON_VALUE)[:200000]
                assert np.array_equal(np.minimum(oc2.query(canon), 255), gc.query(canon))
        finally:
            os.environ.pop("KMU_COUNT_PATH", None)
    wk, wc = oc.dump(2)
    for path in res:
        assert np.array_equal(res[path][0], wk) and np.array_equal(res[path][1], wc), path


def test_count_reference_kat(ctx, oracle):
    """kmercount.rs:1524-1559 / 1580-1617 through the GPU counter"""
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
    s = kat["seq80"]
    bases, off = oracle.concat([s.encode()])
    vk = oracle.kmer_hashes(bases, off, A.KMER16B32BIT, 16, A.FHASH_IDENTITY_RAW)[:65]
    rng = np.random.default_rng(7)
    c = ctx.counter(A.KMER16B32BIT, 16, 8, 1024)
    c.add_kmers(vk[:2].copy())
    c.add_kmers(vk[rng.integers(2, 65, 1_000_000)])
    c.add_kmers(vk[1:2].copy())
    assert c.query(vk[:2].copy()).tolist() == [1, 2]
    c2 = ctx.counter(A.KMER16B32BIT, 16, 8, 1024)
    c2.add_kmers(vk[rng.integers(32, 65, 1_000_000)])
    got = c2.query(vk.copy())
    assert (got[:32] == 0).all() and (got[32:] == 255).all()


def test_full_size_properties(ctx, monkeypatch):
    """size-independent checks at a larger scale than the oracle is run at: determinism, independence of the route (one
    kernel / multiset + points kernels / general instantiation), strand invariance of the canonical sketch, and count
    conservation (sum of counts == number of k-mers)."""
    import torch
    dev = torch.device("cuda:0")
    bases, off, lens = synth.ont_reads_device(20000, 20000 * 6000, 5_000_000, 0xC3, dev)
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 200, A.SIG_U64, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    s1 = ctx.sketch(bases, off, p)
    s2 = ctx.sketch(bases, off, p)
    assert torch.equal(s1, s2)
    for env in ({"KMU_PMH_SPLIT": "1"}, {"KMU_PMH_SPLIT": "0"}, {"KMU_PMH_SPLIT": "0", "KMU_PMH_PLAIN": "0"},
                {"KMU_PMH_SPLIT": "1", "KMU_PMH_PLAIN": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        assert torch.equal(ctx.sketch(bases, off, p), s1), env
        for k_ in env:
            monkeypatch.delenv(k_)
    # reverse-complement every read (on the device): canonical hashing => identical signatures
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    n = len(lens)
    rid = torch.repeat_interleave(torch.arange(n, device=dev), torch.from_numpy(lens).to(dev))
    pos = torch.arange(bases.numel(), device=dev) - off[rid]
    src = off[rid] + (torch.from_numpy(lens).to(dev)[rid] - 1 - pos)
    rc = comp[bases[src].long()]
    rcb = torch.empty(bases.numel() + 64, dtype=torch.uint8, device=dev)
    rcb[:bases.numel()] = rc
    s3 = ctx.sketch(rcb[:bases.numel()], off, p)
    assert torch.equal(s1, s3)
    nk = int(np.maximum(lens - 30, 0).sum())
    c = ctx.counter(A.KMER64BIT, 31, 16, nk)
    c.add_reads(bases, off)
    k_all, c_all = c.export_part(0, 1)
    assert int(c_all.astype(np.int64).sum()) == nk
    assert k_all.size == c.nb_distinct()


# ---- signature comparison ------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,m", [(np.uint32, 200), (np.uint64, 200), (np.float64, 64), (np.float32, 37)])
def test_sig_equal_pairs_and_matrix(ctx, oracle, dtype, m):
    rng = np.random.default_rng(11)
    na, nb = 150, 97
    # few distinct values per slot so that equal slots are common
    a = rng.integers(0, 3, size=(na, m)).astype(dtype)
    b = rng.integers(0, 3, size=(nb, m)).astype(dtype)
    ia = rng.integers(0, na, size=500).astype(np.uint32)
    ib = rng.integers(0, nb, size=500).astype(np.uint32)
    want = np.array([oracle.sig_equal_count(a[i], b[j]) for i, j in zip(ia, ib)], np.uint32)
    got = ctx.sig_equal_pairs(a, b, ia, ib)
    assert np.array_equal(got, want)
    mat = ctx.sig_equal_matrix(a, b)
    assert mat.shape == (na, nb)
    want_m = (a[:, None, :] == b[None, :, :]).sum(-1).astype(np.uint16)
    assert np.array_equal(mat, want_m)
    # device-resident rows give the same answer
    import torch
    ta, tb = torch.from_numpy(a.view(np.int32 if a.itemsize == 4 else np.int64)).cuda(), \
        torch.from_numpy(b.view(np.int32 if b.itemsize == 4 else np.int64)).cuda()
    g2 = ctx.sig_equal_pairs(ta, tb, torch.from_numpy(ia.view(np.int32)).cuda(), torch.from_numpy(ib.view(np.int32)).cuda())
    ctx.synchronize()
    assert np.array_equal(g2.cpu().numpy().view(np.uint32), want)
    m2 = ctx.sig_equal_matrix(ta, tb)
    ctx.synchronize()
    assert np.array_equal(m2.cpu().numpy().view(np.uint16), want_m)


@pytest.mark.gpu
def test_sig_equal_on_real_signatures(ctx):
    """J(read, revcomp(read)) == 1 with the canonical closure (the reference's own check, seqsketchjaccard.rs:785,903)
    and the block distance rule of DistBlockSketched (1.0 inside a sequence)."""
    from kmerutils_amd import sketching as S
    rng = np.random.default_rng(5)
    reads = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n)) for n in (3000, 4000, 2500)]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rc = [r.translate(comp)[::-1] for r in reads]
    sk = S.ProbHash3aSketch(S.SeqSketcherParams(16, 200), kmer_type=A.KMER16B32BIT, ctx=ctx)
    sa = np.asarray(sk.sketch_compressedkmer(reads, A.FHASH_CANON_INVHASH))
    sb = np.asarray(sk.sketch_compressedkmer(rc, A.FHASH_CANON_INVHASH))
    jm = S.jaccard_matrix(sa, sb, ctx=ctx)
    assert np.allclose(np.diag(jm), 1.0)
    assert jm[0, 1] < 0.1
    jp, common = S.probminhash_get_jaccard_objects(sa[0], sb[0], ctx=ctx)
    assert jp == 1.0 and len(common) == 200
    bs = S.BlockSeqSketcher(1000, 12, 64, ctx=ctx)
    rows, numseq, _ = bs.blocksketch_sequences(reads, A.FHASH_CANON_INVHASH)
    rows = np.asarray(rows)
    ia = np.array([0, 0, 1], np.uint32)
    ib = np.array([1, len(rows) - 1, 1], np.uint32)
    d = S.DistBlockSketched(ctx).eval_pairs(rows, numseq, ia, ib)
    assert d[0] == 1.0 and d[2] == 1.0           # same sequence
    want = (rows[0] != rows[-1]).sum() / np.float32(64)
    assert abs(d[1] - want) < 1e-6


@pytest.mark.gpu
def test_minhash_distance_pairs(ctx, oracle):
    rng = np.random.default_rng(3)
    m, n = 50, 40
    rows = np.full((n, m), np.uint64(0xFFFFFFFFFFFFFFFF))
    lens = rng.integers(0, m + 1, size=n)
    lens[0], lens[1] = m, 0
    pool = rng.integers(1, 400, size=4000).astype(np.uint64)
    for i in range(n):
        rows[i, :lens[i]] = np.sort(rng.choice(np.unique(pool), size=lens[i], replace=False))
    ia = rng.integers(0, n, size=300).astype(np.uint32)
    ib = rng.integers(0, n, size=300).astype(np.uint32)
    ia[:3], ib[:3] = [0, 1, 0], [0, 0, 1]
    got = np.asarray(ctx.minhash_distance_pairs(rows, rows, ia, ib))
    want = np.array([oracle.minhash_distance(rows[i, :lens[i]], rows[j, :lens[j]]) for i, j in zip(ia, ib)], np.uint32)
    assert np.array_equal(got, want)
    assert tuple(got[0]) == (m, m, m)  # a sketch against itself


# ---- ingest ----------------------------------------------------------------------------------------
def _make_fastq(rng, n_reads, crlf=False, final_newline=True, allow_empty=True):
    nl = b"\r\n" if crlf else b"\n"
    recs = []
    for i in range(n_reads):
        L = int(rng.integers(0, 1) if (i == 7 and allow_empty) else rng.integers(1, 40000 if i % 50 == 0 else 3000))
        seq = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), size=L).tobytes()
        if i % 9 == 4 and L:   # a read the reference drops
            pos = int(rng.integers(0, L))
            seq = seq[:pos] + bytes([int(rng.choice(np.frombuffer(b"NnRY-.", np.uint8)))]) + seq[pos + 1:]
        qual = rng.choice(np.frombuffer(b"@+IJ#5", np.uint8), size=L).tobytes()  # '@' and '+' may open a quality line
        recs.append(b"@read%d some text" % i + nl + seq + nl + b"+" + (b"read%d" % i if i % 3 == 0 else b"") + nl + qual)
    return nl.join(recs) + (nl if final_newline else b"")


@pytest.mark.gpu
@pytest.mark.parametrize("crlf,final_newline,n_reads", [(False, True, 300), (True, True, 120), (False, False, 64),
                                                         (False, True, 1)])
def test_ingest_fastq_parity(ctx, oracle, crlf, final_newline, n_reads):
    rng = np.random.default_rng(100 + n_reads)
    fq = _make_fastq(rng, n_reads, crlf, final_newline)
    wb, wo, winfo, widx = oracle.ingest_fastq(fq)
    bases, offs, info, idx = ctx.ingest_fastq(fq, want_index=True)
    got = {k: int(getattr(info, k)) for k in winfo}
    assert got == winfo
    assert np.array_equal(offs, wo) and np.array_equal(idx, widx) and bytes(bases) == bytes(wb)
    # device-resident text, outputs stay on the device and feed the sketcher directly
    import torch
    t = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda()
    b2, o2, info2, i2 = ctx.ingest_fastq(t, want_index=True)
    ctx.synchronize()
    assert bytes(b2.cpu().numpy()) == bytes(wb) and np.array_equal(o2.cpu().numpy().astype(np.uint64), wo)
    assert np.array_equal(i2.cpu().numpy().view(np.uint32), widx)


@pytest.mark.gpu
def test_ingest_fastq_many_short_records(ctx, oracle):
    """70 000 records: record-level scans beyond one workgroup's single pass (tiled scan), reads shorter than a chunk"""
    rng = np.random.default_rng(70)
    L = rng.integers(1, 40, size=70_000)
    seqs = [rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=int(n), p=[0.2495, 0.2495, 0.2495, 0.2495, 0.002]).tobytes()
            for n in L]
    fq = b"".join(b"@r\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" for s in seqs)
    wb, wo, winfo, widx = oracle.ingest_fastq(fq)
    assert 2000 < winfo["nb_bad_reads"] < 6000
    bases, offs, info, idx = ctx.ingest_fastx(fq, want_index=True)
    assert {k: int(getattr(info, k)) for k in winfo} == winfo
    assert np.array_equal(offs, wo) and np.array_equal(idx, widx) and bytes(bases) == bytes(wb)


@pytest.mark.gpu
def test_ingest_fastq_errors(ctx):
    from kmerutils_amd.lib import KmuError
    for broken in (b"@r0\nACGT\n+\n", b"@r0\nACGT\n", b"r0\nACGT\n+\nIIII\n", b"@r0\nACGT\n-\nIIII\n",
                   b"@r0\nACGT\n+\nIIII\n\n", b"@r0\nACGT\n+\nIIII\nr1\nACGT\n+\nIIII\n"):
        with pytest.raises(KmuError) as e:
            ctx.ingest_fastq(broken)
        assert e.value.code == A.E_BAD_ARG
    b, o, info = ctx.ingest_fastq(b"")
    assert info.n_records == 0 and o.tolist() == [0]


def _make_fasta(rng, n_records, width, crlf=False, final_newline=True, allow_empty=True):
    """FASTA text: records of 0 .. 40 k bases wrapped at `width` columns (0 = one line per record), some with N, some
    in lower case, headers that contain '>' and blanks, an occasional empty line"""
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for r in range(n_records):
        L = int(rng.choice([0, 1, 5, 59, 60, 61, 500, 16384, 40000])) if r % 3 == 0 else int(rng.integers(1, 3000))
        L = max(L, 0 if allow_empty else 1)
        seq = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), size=L).tobytes()
        if r % 7 == 5 and L > 2:
            seq = seq[:L // 2] + b"N" + seq[L // 2 + 1:]
        out.append(b">rec%d some>thing here" % r)
        if width and L:
            out += [seq[i:i + width] for i in range(0, L, width)]
        else:
            out.append(seq)
        if r % 11 == 2:
            out.append(b"")
    text = nl.join(out)
    return text + nl if final_newline else text


@pytest.mark.gpu
@pytest.mark.parametrize("width,crlf,final_newline,n", [(60, False, True, 150), (1, False, True, 12), (80, True, True, 90),
                                                       (0, False, False, 40), (17, True, False, 60), (70, False, True, 1),
                                                       (2, False, True, 120)])  # > 2^15 lines: the tiled scans
def test_ingest_fasta_parity(ctx, oracle, width, crlf, final_newline, n):
    """kmu_ingest_fasta / kmu_ingest_fastx against the oracle's needletail-style reader: multi-line records, any line
    width (single-line records longer than a walk region included), CRLF, empty lines, no final newline"""
    import torch
    rng = np.random.default_rng(1000 + 7 * width + n)
    fa = _make_fasta(rng, n, width, crlf, final_newline)
    wb, wo, winfo, widx = oracle.ingest_fasta(fa)
    assert winfo["n_records"] == n
    bases, offs, info, idx = ctx.ingest_fasta(fa, want_index=True)
    assert {k: int(getattr(info, k)) for k in winfo} == winfo
    assert np.array_equal(offs, wo) and np.array_equal(idx, widx) and bytes(bases) == bytes(wb)
    b1, o1, info1 = ctx.ingest_fastx(fa)  # format from the first byte
    assert bytes(b1) == bytes(wb) and np.array_equal(o1, wo)
    t = torch.from_numpy(np.frombuffer(fa, np.uint8).copy()).cuda()
    b2, o2, info2, i2 = ctx.ingest_fastx(t, want_index=True)
    ctx.synchronize()
    assert bytes(b2.cpu().numpy()) == bytes(wb) and np.array_equal(o2.cpu().numpy().astype(np.uint64), wo)
    assert np.array_equal(i2.cpu().numpy().view(np.uint32), widx)


@pytest.mark.gpu
def test_ingest_fastx_dispatch_and_errors(ctx, oracle):
    from kmerutils_amd.lib import KmuError
    fq = b"@r0\nACGT\n+\nIIII\n@r1\nACNT\n+\nIIII\n"
    b, o, info = ctx.ingest_fastx(fq)
    assert bytes(b) == b"ACGT" and o.tolist() == [0, 4] and info.nb_bad_reads == 1
    for broken in (b"ACGT\n>r0\nACGT\n", b"\n>r0\nACGT\n", b"#comment\n"):
        with pytest.raises(KmuError) as e:
            ctx.ingest_fastx(broken)
        assert e.value.code == A.E_BAD_ARG
        with pytest.raises(oracle.OracleError):
            oracle.ingest_fastx(broken)
    with pytest.raises(KmuError):
        ctx.ingest_fasta(b"ACGT\n>r0\nACGT\n")
    # a genome the way gsearch sees it: contigs of one FASTA file -> ONE signature (sketch_compressedkmer_seqs)
    rng = np.random.default_rng(5)
    fa = _make_fasta(rng, 40, 60, allow_empty=False)
    bases, offs, info = ctx.ingest_fastx(fa)
    wb, wo, _, _ = oracle.ingest_fastx(fa)
    keep = np.diff(wo.astype(np.int64)) >= 21  # contigs shorter than k contribute nothing
    assert keep.sum() > 10
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 21, 128, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0,
                       A.MODE_ALL_SEQS, 0, 0, 0)
    assert np.array_equal(np.asarray(ctx.sketch(bases, offs, p)), oracle.sketch(wb, wo, p))


@pytest.mark.gpu
def test_ingest_then_sketch_equals_direct(ctx, oracle):
    """end to end on the device: FASTQ text -> kmu_ingest_fastq -> kmu_sketch, against the oracle reader + oracle sketch"""
    import torch
    rng = np.random.default_rng(77)
    fq = _make_fastq(rng, 200, allow_empty=False)
    t = torch.from_numpy(np.frombuffer(fq, np.uint8).copy()).cuda()
    bases, offs, info = ctx.ingest_fastq(t)
    assert info.n_kept > 150 and info.nb_bad_reads > 10
    wb, wo, _, _ = oracle.ingest_fastq(fq)
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 100, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    want = oracle.sketch(wb, wo, p)
    got = ctx.sketch(bases, offs, p)
    ctx.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint64), want)


@pytest.mark.gpu
def test_probminhash3_depth_first_oracle(ctx, oracle):
    """KMU_ALGO_PROB3 (sketch_probminhash3, seqsketchjaccard.rs:272-319): the device runs the ProbMinHash3a kernel; the
    oracle's key-by-key (depth-first) formulation is an independent check that both keep the same per-slot minimum."""
    rng = np.random.default_rng(9)
    seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(n)).tobytes() for n in (40, 900, 5000, 23000)]
    seqs.append(b"ACGTTGCA" * 400)  # heavy multiplicities
    bases, off = oracle.concat(seqs)
    for kmer_type, k, sig in ((A.KMER32BIT, 7, A.SIG_U32), (A.KMER64BIT, 25, A.SIG_U64)):
        p3 = A.SketchParams(A.ALGO_PROB3, kmer_type, k, 150, sig, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
        p3a = A.SketchParams(A.ALGO_PROB3A, kmer_type, k, 150, sig, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
        want3 = oracle.sketch(bases, off, p3)
        assert np.array_equal(want3, oracle.sketch(bases, off, p3a))  # the two oracle formulations agree
        assert np.array_equal(np.asarray(ctx.sketch(bases, off, p3)), want3)
    from kmerutils_amd import sketching as S
    sk = S.SeqSketcher(25, 150, ctx=ctx)
    a = np.asarray(sk.sketch_probminhash3(seqs, A.FHASH_CANON_INVHASH))
    assert np.array_equal(a, want3)


@pytest.mark.gpu
def test_two_kernel_probminhash_path(ctx, oracle, monkeypatch):
    """KMU_PMH_SPLIT=1: multiset kernel -> (key, weight) lists -> k_pmh_points; same rows as the oracle, including
    reads that need several partition passes, tandem repeats and a read shorter than k."""
    monkeypatch.setenv("KMU_PMH_SPLIT", "1")
    monkeypatch.setenv("KMU_PMH_SMALLK", "0")  # (k = 8 has a route of its own: test_smallk_histogram_route)
    rng = np.random.default_rng(21)
    seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(n)).tobytes() for n in (12, 300, 7000, 45000, 9000)]
    seqs.append(b"ACGGT" * 3000)
    seqs.append(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=900).tobytes() * 9)  # every k-mer nine times: all keys in collision groups
    seqs.append(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=4000).tobytes() + rng.choice(np.frombuffer(b"ACGT", np.uint8), size=150).tobytes() * 20)
    seqs.append(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=6000).tobytes() + rng.choice(np.frombuffer(b"ACGT", np.uint8), size=100).tobytes() * 3)  # a few hundred keys of weight 3 among the false positives: the groups the counting sort merges
    bases, off = oracle.concat(seqs)
    for kmer_type, k, sig, m in ((A.KMER64BIT, 31, A.SIG_U64, 200), (A.KMER32BIT, 8, A.SIG_U32, 64)):
        p = A.SketchParams(A.ALGO_PROB3A, kmer_type, k, m, sig, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
        ctx.profile_reset()
        ctx.profile_enable(True)
        got = np.asarray(ctx.sketch(bases, off, p))
        ctx.profile_enable(False)
        assert "k_pmh_points" in ctx.profile_get()
        assert np.array_equal(got, oracle.sketch(bases, off, p))


@pytest.mark.gpu
def test_io_mirror_parse_and_blocks(ctx, oracle, tmp_path):
    """kmerutils_amd.io: parse_with_needletail (src/io.rs:12-72) on a file, readblockseq batching
    (datasketcher.rs:358-388) over the accepted reads"""
    from kmerutils_amd import io as kio
    rng = np.random.default_rng(12)
    fq = _make_fastq(rng, 90, allow_empty=False)
    fn = tmp_path / "reads.fastq"
    fn.write_bytes(fq)
    bases, offs, stats = kio.parse_with_needletail(str(fn), ctx=ctx)
    wb, wo, winfo, _ = oracle.ingest_fastq(fq)
    assert bytes(bases) == bytes(wb) and np.array_equal(offs, wo)
    assert stats == dict(nb_rec_loaded=winfo["n_kept"], nb_bases=winfo["n_bases"], nb_bad_bases=winfo["nb_bad_bases"],
                         nb_bad_read=winfo["nb_bad_reads"], nb_records=winfo["n_records"])
    got = []
    first = 0
    while first < len(offs) - 1:   # blocks of 25 reads, like the pack-of-10 000 loop of datasketcher
        b, o = kio.readblockseq(bases, offs, first, 25)
        got += [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(len(o) - 1)]
        first += len(o) - 1
    assert got == [bytes(wb[int(wo[i]):int(wo[i + 1])]) for i in range(len(wo) - 1)]


@pytest.mark.gpu
def test_datasketcher_tool_end_to_end(oracle, tmp_path):
    """the datasketcher mirror: FASTQ file -> device ingest -> ProbMinHash3a -> the reference's dump format"""
    from kmerutils_amd import datasketcher, formats
    rng = np.random.default_rng(31)
    fq = _make_fastq(rng, 60, allow_empty=False)
    fn, dump = tmp_path / "r.fastq", tmp_path / "r.sig"
    fn.write_bytes(fq)
    assert datasketcher.main(["-f", str(fn), "-k", "8", "-s", "64", "-d", str(dump)]) == 0
    wb, wo, _, _ = oracle.ingest_fastq(fq)
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER32BIT, 8, 64, A.SIG_U32, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    want = oracle.sketch(wb, wo, p)
    r = formats.SigSketchFileReader(str(dump))
    assert (r.get_kmer_size(), r.get_signature_length()) == (8, 64)
    assert np.array_equal(r.read_all(), want)
    # by blocks of 1000 bases
    bdump = tmp_path / "r.blk"
    assert datasketcher.main(["-f", str(fn), "-k", "8", "-s", "32", "-d", str(bdump), "-b", "1000"]) == 0
    pb = A.SketchParams(A.ALGO_PROB3A, A.KMER32BIT, 8, 32, A.SIG_U32, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 1000, 0, 0, 0, 0)
    wantb = oracle.sketch(wb, wo, pb)
    rb = formats.SigBlockSketchFileReader(str(bdump))
    rows = []
    seq = 0
    while True:
        nxt = rb.next()
        if nxt is None:
            break
        assert nxt[0] == seq
        rows += [b[1] for b in nxt[1]]
        seq += 1
    assert seq == len(wo) - 1 and np.array_equal(np.array(rows), wantb)


@pytest.mark.gpu
def test_parsefastq_tool_end_to_end(oracle, tmp_path):
    """the parsefastq mirror (counting branch): FASTQ file -> device ingest -> counts -> COUNTER_MULTIPLE dump"""
    from kmerutils_amd import formats, parsefastq
    rng = np.random.default_rng(41)
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=3000).tobytes()
    recs = []
    for i in range(80):  # overlapping reads from one genome: many k-mers seen more than once
        s = int(rng.integers(0, 2500))
        seq = genome[s:s + int(rng.integers(100, 500))]
        if i % 10 == 3:
            seq = seq[:5] + b"N" + seq[6:]
        recs.append(b"@r%d\n" % i + seq + b"\n+\n" + b"I" * len(seq))
    fn = tmp_path / "g.fastq"
    fn.write_bytes(b"\n".join(recs) + b"\n")
    for k, kmer_type, vb in ((21, A.KMER64BIT, 8), (16, A.KMER16B32BIT, 4), (11, A.KMER32BIT, 4)):
        assert parsefastq.main(["-f", str(fn), "-s", str(k), "--outdir", str(tmp_path)]) == 0
        wb, wo, _, _ = oracle.ingest_fastq(fn.read_bytes())
        oc = oracle.Counter(kmer_type, k, 8, 1 << 20)
        oc.add_reads(wb, wo)
        wk, wc = oc.dump(2)
        ks, vals, cnts = formats.load_kmer_counter(str(tmp_path / "g.fastq.multi_kmer.bin"), vb)
        order = np.argsort(vals)
        assert ks == k and np.array_equal(vals[order], wk) and np.array_equal(cnts[order].astype(np.uint32), np.minimum(wc, 255))


@pytest.mark.gpu
def test_probminhash_megabase_read(ctx, oracle):
    """a 1.3 Mbase sequence next to an ordinary read (since the genome-sized route exists it is sketched there; the
    in-LDS kernel keeps sequences up to 2^18 k-mers, i.e. up to ~25 partition passes)"""
    rng = np.random.default_rng(55)
    seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=1_300_000).tobytes(),
            rng.choice(np.frombuffer(b"ACGT", np.uint8), size=25_000).tobytes()]
    bases, off = oracle.concat(seqs)
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 64, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    assert np.array_equal(np.asarray(ctx.sketch(bases, off, p)), oracle.sketch(bases, off, p))


@pytest.mark.gpu
def test_probminhash_genome_sized_sequences(ctx, oracle):
    """sequences far beyond one LDS pass (genomes rather than reads) take the global partitioned route inside
    kmu_sketch; rows of ordinary reads in the same call are unaffected; device-resident input as well"""
    import time
    import torch
    rng = np.random.default_rng(56)
    lens = (3_000, 2_600_000, 40_000, 700_000, 500)
    seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n).tobytes() for n in lens]
    bases, off = oracle.concat(seqs)
    for kmer_type, k, sig in ((A.KMER64BIT, 21, A.SIG_U64), (A.KMER32BIT, 12, A.SIG_U32)):
        p = A.SketchParams(A.ALGO_PROB3A, kmer_type, k, 128, sig, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
        want = oracle.sketch(bases, off, p)
        t0 = time.perf_counter()
        got = np.asarray(ctx.sketch(bases, off, p))
        dt = time.perf_counter() - t0
        assert np.array_equal(got, want)
        assert dt < 2.0, "a 2.6 Mbase sequence must not take L / cap passes (%.2f s)" % dt
        db, do = torch.from_numpy(bases.copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
        g2 = ctx.sketch(db, do, p)
        ctx.synchronize()
        assert np.array_equal(g2.cpu().numpy().view(want.dtype), want)
        # the same sequences packed as Sequence::new(raw, 2) (what the C++ mirror's Sequence hands over)
        packed, poff = ctx.pack2b(bases, off)
        pp = A.SketchParams.from_buffer_copy(p)
        pp.input_kind = A.INPUT_PACKED2
        assert np.array_equal(np.asarray(ctx.sketch(packed, off, pp, packed_offsets=poff)), want)
    # other sketches of a genome-sized sequence stay on their own kernels: SuperMinHash, bottom-k
    ps = A.SketchParams(A.ALGO_SUPER, A.KMER64BIT, 21, 64, A.SIG_F64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    assert np.array_equal(np.asarray(ctx.sketch(bases, off, ps)), oracle.sketch(bases, off, ps))
    pb = A.SketchParams(A.ALGO_BOTTOMK, A.KMER64BIT, 21, 100, A.SIG_U64, A.HASHER_INT64HASH, A.FHASH_CANON_VALUE, 0, 0, 0, 0, 0)
    gh, gc = ctx.sketch(bases, off, pb, want_counts=True)
    wh, wc = oracle.sketch(bases, off, pb, want_counts=True)
    assert np.array_equal(np.asarray(gh), wh) and np.array_equal(np.asarray(gc), wc)


@pytest.mark.gpu
def test_device_buffer_helpers(ctx, oracle):
    """kmu_dev_alloc / kmu_copy_to_device / kmu_copy_to_host / kmu_dev_free: a host without a HIP binding uploads its reads
    once and calls the KMU_MEM_DEVICE entry points on the returned pointers"""
    import ctypes as C
    L = ctx.L
    seqs = ragged_dna(12, RAGGED)
    bases, off = oracle.concat(seqs)
    nb, n = int(off[-1]), len(off) - 1
    d_bases, d_off, d_sig = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for ptr, size in ((d_bases, nb + 64), (d_off, 8 * (n + 1)), (d_sig, n * 32 * 8)):
        assert L.kmu_dev_alloc(ctx.h, size, C.byref(ptr)) == 0 and ptr.value
    assert L.kmu_copy_to_device(ctx.h, d_bases, bases.ctypes.data_as(C.c_void_p), nb) == 0
    assert L.kmu_copy_to_device(ctx.h, d_off, off.ctypes.data_as(C.c_void_p), 8 * (n + 1)) == 0
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 21, 32, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0,
                       A.INPUT_ASCII, A.MEM_DEVICE, 0)
    assert L.kmu_sketch(ctx.h, C.byref(p), d_bases, d_off, None, n, None, d_sig, None) == 0
    got = np.zeros((n, 32), np.uint64)
    assert L.kmu_copy_to_host(ctx.h, got.ctypes.data_as(C.c_void_p), d_sig, got.nbytes) == 0
    p.mem = A.MEM_HOST
    assert np.array_equal(got, oracle.sketch(bases, off, p))
    for ptr in (d_bases, d_off, d_sig):
        assert L.kmu_dev_free(ctx.h, ptr) == 0
    assert L.kmu_dev_free(ctx.h, None) == 0


@pytest.mark.gpu
def test_seqrange_sketches_of_seqminhash(ctx, oracle):
    """kmerutils_amd.sketching.sketch_seqrange_minhash / _superminhash (src/sketching/seqminhash.rs:19-119) with the
    reference's tests :127-258 (two overlapping ranges of the 80-base string; `total == 20` is exact)"""
    from kmerutils_amd import sketching as S
    seq = b"TCAAAGGGAAACATTCAAAATCAGTATGCGCCCGTTCAGTTACGTATTGCTCTCGCTAATGAGATGGGCTGGGTACAGAG"
    for k, check in ((16, lambda t: t >= 3), (10, lambda t: t == 20)):
        h1, c1 = S.sketch_seqrange_minhash(seq, (1, 65), k, 20, ctx=ctx)
        h2, c2 = S.sketch_seqrange_minhash(seq, (35, 75), k, 20, ctx=ctx)
        cont, jac, common, total = S.minhash_distance(h1[None, :], h2[None, :], [0], [0], ctx=ctx)
        assert check(int(total[0])) and int(common[0]) == oracle.minhash_distance(h1, h2)[0]
    for k, m, thresh in ((16, 50, 0.15), (10, 20, 0.2)):
        s1 = S.sketch_seqrange_superminhash(seq, (1, 65), k, m, ctx=ctx)
        s2 = S.sketch_seqrange_superminhash(seq, (35, 75), k, m, ctx=ctx)
        assert float((s1 == s2).mean()) >= thresh
    with pytest.raises(ValueError):
        S.sketch_seqrange_superminhash(seq, (1, 65), 8, 20, ctx=ctx)


@pytest.mark.gpu
def test_ranges_of_a_larger_read_set(ctx, oracle, monkeypatch):
    """offsets[0] > 0: a call may name a range of reads of a larger array (`offsets + first`, what the C++ tools do with
    device-resident reads).  Counting (direct and partitioned paths, owner grouping), sketching, k-mer hashes, packing and
    the non-ACGT census see exactly the reads of the range, for host and for device input."""
    import torch
    rng = np.random.default_rng(3)
    seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(n)).tobytes() for n in rng.integers(30, 3000, size=400)]
    bases, off = oracle.concat(seqs)
    db, do = torch.from_numpy(bases.copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 21, 64, A.SIG_U64, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    for first, last in ((0, 400), (37, 211), (399, 400), (150, 400)):
        sub_b, sub_o = oracle.concat(seqs[first:last])
        oc = oracle.Counter(A.KMER64BIT, 21, 8, 1 << 20)
        oc.add_reads(sub_b, sub_o)
        wk, wc = oc.dump(1)
        for path in ("partitioned", "direct"):
            monkeypatch.setenv("KMU_COUNT_PATH", path)
            for dev in (False, True):
                c = ctx.counter(A.KMER64BIT, 21, 8, 1 << 20)
                if dev:
                    c.add_reads(db, do[first:last + 1])
                else:
                    c.add_reads(bases, off[first:last + 1].copy())
                gk, gc = c.dump(1)
                assert np.array_equal(gk, wk) and np.array_equal(gc, wc), (first, last, path, dev)
                c.close()
        monkeypatch.delenv("KMU_COUNT_PATH")
        c = ctx.counter(A.KMER64BIT, 21, 8, 1 << 20)
        kmers, bounds = c.extract_by_owner(db, do[first:last + 1], 3)
        canon = oracle.kmer_hashes(sub_b, sub_o, A.KMER64BIT, 21, A.FHASH_CANON_VALUE)
        valid = np.concatenate([np.arange(int(sub_o[i]), int(sub_o[i + 1]) - 20) for i in range(last - first)
                                if sub_o[i + 1] - sub_o[i] >= 21])
        assert int(bounds[-1]) == valid.size
        assert np.array_equal(np.sort(kmers.cpu().numpy().view(np.uint64)), np.sort(canon[valid]))
        c.close()
        want = oracle.sketch(sub_b, sub_o, p)
        assert np.array_equal(np.asarray(ctx.sketch(bases, off[first:last + 1].copy(), p)), want)
        for split in ("0", "1"):  # one kernel / the two-kernel route of big batches (its lists are indexed from the range's start)
            monkeypatch.setenv("KMU_PMH_SPLIT", split)
            g = ctx.sketch(db, do[first:last + 1], p)
            ctx.synchronize()
            assert np.array_equal(g.cpu().numpy().view(np.uint64), want)
        monkeypatch.delenv("KMU_PMH_SPLIT")
        for algo, sig, mode in ((A.ALGO_SUPER, A.SIG_F64, A.MODE_PER_SEQ), (A.ALGO_OPTDENS, A.SIG_F64, A.MODE_PER_SEQ),
                                (A.ALGO_REVOPTDENS, A.SIG_F32, A.MODE_ALL_SEQS), (A.ALGO_PROB3A, A.SIG_U64, A.MODE_ALL_SEQS),
                                (A.ALGO_SUPER2, A.SIG_U64, A.MODE_ALL_SEQS), (A.ALGO_BOTTOMK, A.SIG_U64, A.MODE_PER_SEQ)):
            pa = A.SketchParams(algo, A.KMER64BIT, 21, 48, sig, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, mode, 0, 0, 0)
            wa = oracle.sketch(sub_b, sub_o, pa)
            ga = ctx.sketch(db, do[first:last + 1], pa)
            ctx.synchronize()
            assert np.array_equal(ga.cpu().numpy().view(np.uint8), np.ascontiguousarray(wa).view(np.uint8)), (algo, mode)
            gh = np.asarray(ctx.sketch(bases, off[first:last + 1].copy(), pa))
            assert np.array_equal(np.ascontiguousarray(gh).view(np.uint8), np.ascontiguousarray(wa).view(np.uint8)), (algo, mode)
        # per-position hashes land at the caller's absolute positions
        hk = ctx.kmer_hashes(bases, off[first:last + 1].copy(), A.KMER64BIT, 21, A.FHASH_CANON_INVHASH,
                             out=np.zeros(int(off[-1]), np.uint64))
        wh = oracle.kmer_hashes(sub_b, sub_o, A.KMER64BIT, 21, A.FHASH_CANON_INVHASH)
        assert np.array_equal(hk[int(off[first]):int(off[last])], wh[:int(sub_o[-1])])
        assert not hk[:int(off[first])].any() and not hk[int(off[last]):].any()
        # pre-hashed values of a range, per sequence and for all of them
        vals = np.arange(int(off[-1]), dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        for mode in (A.MODE_PER_SEQ, A.MODE_ALL_SEQS):
            ph = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 21, 32, A.SIG_U64, 0, A.FHASH_IDENTITY_RAW, 0, mode, 0, 0, 0)
            wantv = oracle.sketch_hashed(vals[int(off[first]):int(off[last])].copy(), sub_o, ph)
            assert np.array_equal(np.asarray(ctx.sketch_hashed(vals, off[first:last + 1].copy(), ph)), wantv)
        packed, poff = ctx.pack2b(bases, off[first:last + 1].copy())
        assert bytes(packed) == b"".join(bytes(oracle.pack2b(s)) for s in seqs[first:last])
        assert not ctx.count_non_acgt(bases, off[first:last + 1].copy()).any()


@pytest.mark.gpu
def test_once_kmers_with_positions(ctx, oracle, tmp_path):
    """kmu_count_once_positions: the contract of KmerFilter1 (kmercount.rs:985-1082) -- the k-mers seen exactly once in the
    read set, each with (numseq, numkmer), in file order; host and device input, a range of reads, three k-mer types; and
    the COUNTER_UNIQUE dump round trip"""
    import torch
    from kmerutils_amd import formats
    rng = np.random.default_rng(985)
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=20_000).tobytes()
    seqs = []
    for i in range(300):
        L = int(rng.integers(5, 700))
        s0 = int(rng.integers(0, len(genome) - L))
        s = genome[s0:s0 + L]
        seqs.append(s if i % 3 else s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")))  # both strands
    bases, off = oracle.concat(seqs)
    db, do = torch.from_numpy(bases.copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
    for kmer_type, k, vb in ((A.KMER16B32BIT, 16, 4), (A.KMER64BIT, 25, 8), (A.KMER32BIT, 11, 4)):
        c = ctx.counter(kmer_type, k, 8, 1 << 18)
        o = oracle.Counter(kmer_type, k, 8, 1 << 18)
        c.add_reads(bases, off)
        o.add_reads(bases, off)
        wk, ws, wp = o.once_positions(bases, off)
        assert 100 < wk.size < off[-1]
        gk, gs, gp = c.once_positions(bases, off)
        assert np.array_equal(gk, wk) and np.array_equal(gs, ws) and np.array_equal(gp, wp)
        dk, dsq, dp = c.once_positions(db, do)
        ctx.synchronize()
        assert np.array_equal(dk.cpu().numpy().view(np.uint64), wk) and np.array_equal(dsq.cpu().numpy().view(np.uint32), ws)
        assert np.array_equal(dp.cpu().numpy().view(np.uint32), wp)
        # every reported k-mer is unique in the table; the positions of a range of reads are relative to the range
        assert (c.query(gk) == 1).all() and o.nb_unique() == np.unique(wk).size
        rk, rs, rp = c.once_positions(db, do[100:181])
        sel = (ws >= 100) & (ws < 180)
        assert np.array_equal(rk.cpu().numpy().view(np.uint64), wk[sel])
        assert np.array_equal(rs.cpu().numpy().view(np.uint32), ws[sel] - 100) and np.array_equal(rp.cpu().numpy().view(np.uint32), wp[sel])
        fn = str(tmp_path / "reads.once_kmer.bin")
        assert formats.dump_once_kmers(fn, gk, gs, gp, k, vb) == gk.size
        k2, vk, vs, vp = formats.load_once_kmers(fn, vb)
        assert k2 == k and np.array_equal(vk, wk) and np.array_equal(vs, ws) and np.array_equal(vp, wp)
        c.close()
    # nothing counted yet: no record
    c = ctx.counter(A.KMER64BIT, 25, 8, 1 << 12)
    assert c.once_positions(bases, off)[0].size == 0


@pytest.mark.gpu
def test_partial_sketches_merge_to_the_all_sequences_sketch(ctx, oracle):
    """kmu_sketch_partial / kmu_sketch_hashed_partial / kmu_sketch_merge_partials: the signature of a list of sequences
    from the per-slot minima of shares of it -- shares of the sequences for the unweighted sketches, disjoint key sets for
    ProbMinHash -- equals kmu_sketch(ALL_SEQS) of the whole list (and the oracle's); host and device buffers"""
    import torch
    rng = np.random.default_rng(160)
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=40_000).tobytes()
    seqs = []
    for _ in range(90):
        L = int(rng.integers(10, 2500))
        s0 = int(rng.integers(0, len(genome) - L))
        seqs.append(genome[s0:s0 + L])   # overlapping samples: multiplicities well above 1
    bases, off = oracle.concat(seqs)
    shares = [(0, 25), (25, 26), (26, 90)]
    for algo, sig, m in ((A.ALGO_SUPER, A.SIG_F64, 128), (A.ALGO_SUPER, A.SIG_F32, 64), (A.ALGO_SUPER2, A.SIG_U64, 200),
                         (A.ALGO_SUPER2, A.SIG_U32, 64), (A.ALGO_OPTDENS, A.SIG_F64, 3000), (A.ALGO_REVOPTDENS, A.SIG_F32, 500)):
        p = A.SketchParams(algo, A.KMER64BIT, 21, m, sig, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, A.MODE_ALL_SEQS, 0, 0, 0)
        want = np.ascontiguousarray(oracle.sketch(bases, off, p)).view(np.uint8)
        parts = np.stack([np.asarray(ctx.sketch_partial(bases, off[a:b + 1].copy(), p)) for a, b in shares])
        got = np.asarray(ctx.sketch_merge_partials(parts, p))
        assert np.array_equal(np.ascontiguousarray(got).view(np.uint8), want.reshape(-1)), (algo, sig)
        db, do = torch.from_numpy(bases.copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
        dparts = torch.stack([ctx.sketch_partial(db, do[a:b + 1], p) for a, b in shares])
        gd = ctx.sketch_merge_partials(dparts, p)
        ctx.synchronize()
        assert np.array_equal(gd.cpu().numpy().view(np.uint8), want.reshape(-1)), (algo, sig)
    # ProbMinHash: the weight of a key is its multiplicity over ALL sequences -> shares are disjoint key sets
    for kmer_type, k, sig in ((A.KMER64BIT, 21, A.SIG_U64), (A.KMER32BIT, 9, A.SIG_U32)):
        p = A.SketchParams(A.ALGO_PROB3A, kmer_type, k, 150, sig, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, A.MODE_ALL_SEQS, 0, 0, 0)
        want = oracle.sketch(bases, off, p)[0]
        h = oracle.kmer_hashes(bases, off, kmer_type, k, A.FHASH_CANON_INVHASH)
        valid = np.concatenate([np.arange(int(off[i]), int(off[i + 1]) - k + 1) for i in range(len(off) - 1) if off[i + 1] - off[i] >= k])
        vals = h[valid]
        owner = (vals * np.uint64(0x9E3779B97F4A7C15) >> np.uint64(40)) % np.uint64(3)
        dt = np.uint32 if sig == A.SIG_U32 else np.uint64
        parts = []
        for r in range(3):
            mine = vals[owner == r].astype(dt)
            if r == 2:  # one share handed over as two "sequences": still one multiset
                o2 = np.array([0, mine.size // 3, mine.size], np.uint64)
            else:
                o2 = np.array([0, mine.size], np.uint64)
            parts.append(np.asarray(ctx.sketch_hashed_partial(mine, o2, p)))
        parts.append(np.asarray(ctx.sketch_hashed_partial(np.zeros(1, dt), np.array([0, 0], np.uint64), p)))  # a rank without data
        got = np.asarray(ctx.sketch_merge_partials(np.stack(parts), p))
        assert np.array_equal(got, want)
        # and splitting by SEQUENCES instead is (as it must be) not the same thing: weights would be partial
        naive = np.stack([np.asarray(ctx.sketch_partial(bases, off[a:b + 1].copy(), p)) for a, b in shares])
        assert not np.array_equal(np.asarray(ctx.sketch_merge_partials(naive, p)), want)


@pytest.mark.gpu
@pytest.mark.parametrize("thr", ["1024", "0"])
def test_points_of_long_reads_by_workgroups(ctx, oracle, monkeypatch, thr):
    """k_pmh_points hands reads with more than KMU_PMH_PTS_LONG list entries (default 32 768; here 1 024, and 0 = never) to a
    whole workgroup: four waves with slot minima of their own on every fourth chunk of the list, the row their per-slot
    minimum.  Same rows as the oracle -- for unique keys, for weights > 1 (tandem repeats, a unit repeated ten times: the later
    ProbMinHash rounds run per wave against the merged q_max), for lists that are not a multiple of a chunk, both signature
    widths, m below and above a wave."""
    monkeypatch.setenv("KMU_PMH_SPLIT", "1")
    monkeypatch.setenv("KMU_PMH_PTS_LONG", thr)
    rng = np.random.default_rng(2103)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seqs = [rng.choice(acgt, size=int(n)).tobytes() for n in (12, 300, 1100, 1054, 5000, 20001, 45000, 9000, 1311)]
    seqs.append(b"ACGGT" * 3000)
    seqs.append(rng.choice(acgt, size=2000).tobytes() * 10)
    seqs.append(rng.choice(acgt, size=700).tobytes() * 3 + rng.choice(acgt, size=3000).tobytes())
    bases, off = oracle.concat(seqs)
    for kmer_type, k, sig, m in ((A.KMER64BIT, 31, A.SIG_U64, 200), (A.KMER64BIT, 21, A.SIG_U64, 37), (A.KMER16B32BIT, 16, A.SIG_U32, 64)):
        p = A.SketchParams(A.ALGO_PROB3A, kmer_type, k, m, sig, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
        ctx.profile_reset()
        ctx.profile_enable(True)
        got = np.asarray(ctx.sketch(bases, off, p))
        ctx.profile_enable(False)
        assert "k_pmh_points" in ctx.profile_get()
        assert np.array_equal(got, oracle.sketch(bases, off, p)), (kmer_type, k, m)


@pytest.mark.gpu
def test_probminhash_many_reads_default_route(ctx, oracle, monkeypatch):
    """A big batch of reads takes the two-kernel route by itself (multiset kernel -> (key, weight) lists -> k_pmh_points,
    one wave per read): rows equal to the oracle's, for u64 and u32 signatures, with reads shorter than k and repetitive
    reads (weights > 1, later ProbMinHash rounds) in the batch.  A batch whose longest read would keep one wave busy long
    after the others have finished stays on the single kernel."""
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seqs = []
    for i in range(60000):
        if i % 13 == 0:
            unit = rng.choice(acgt, size=int(rng.integers(1, 40))).tobytes()
            seqs.append((unit * 40)[:int(rng.integers(5, 600))])
        else:
            seqs.append(rng.choice(acgt, size=int(rng.integers(1, 450))).tobytes())
    bases, off = oracle.concat(seqs)

    def kernels_of(call):
        ctx.profile_reset()
        ctx.profile_enable(True)
        out = call()
        names = set(ctx.profile_get())
        ctx.profile_enable(False)
        return out, names

    for kmer_type, k, sig in ((A.KMER64BIT, 31, A.SIG_U64), (A.KMER32BIT, 8, A.SIG_U32)):
        p = A.SketchParams(A.ALGO_PROB3A, kmer_type, k, 200, sig, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, A.MODE_PER_SEQ,
                           A.INPUT_ASCII, A.MEM_HOST, 0)
        want = oracle.sketch(bases, off, p)
        got, names = kernels_of(lambda: np.asarray(ctx.sketch(bases, off, p)))
        assert "k_pmh_points" in names, names
        assert got.tobytes() == np.asarray(want).tobytes()
    # the same batch with one 300 kb read in it: the single kernel, same rows for the reads they share
    seqs2 = seqs[:20000] + [rng.choice(acgt, size=300_000).tobytes()]
    b2, o2 = oracle.concat(seqs2)
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 200, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, A.MODE_PER_SEQ,
                       A.INPUT_ASCII, A.MEM_HOST, 0)
    got2, names2 = kernels_of(lambda: np.asarray(ctx.sketch(b2, o2, p)))
    assert "k_pmh_points" not in names2, names2
    want2 = oracle.sketch(b2, o2, p)
    assert got2.tobytes() == np.asarray(want2).tobytes()


@pytest.mark.parametrize("split", ["1", "0"])
def test_smallk_histogram_route(ctx, oracle, monkeypatch, split):
    """k <= 8: the multiset as a direct-indexed histogram in LDS (k_sketch_smallk), with the points made by k_pmh_points
    (KMU_PMH_SPLIT=1) or by the histogram kernel itself (=0).  Reads of every regime: shorter than k, listed first touches
    (<= 8192 k-mers), scanned histogram, 32-bit counters in two halves (> 65535 k-mers), poly-A (one counter takes it all),
    tandem repeats; every closure the route accepts; packed input; against the oracle and against the general kernels."""
    monkeypatch.setenv("KMU_PMH_SPLIT", split)
    rng = np.random.default_rng(88)
    seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(n)).tobytes() for n in (5, 8, 9, 300, 8199, 8200, 30000, 65542, 65543, 150000)]
    seqs += [b"A" * 70000, b"ACGGT" * 2000, b"AC" * 40000, b"T" * 9]
    bases, off = oracle.concat(seqs)
    packed, poff = ctx.pack2b(bases, off)
    for k, fh, m in ((8, A.FHASH_CANON_INVHASH, 200), (8, A.FHASH_IDENTITY_RAW, 64), (5, A.FHASH_CANON_VALUE, 33), (3, A.FHASH_INVHASH_RAW, 16),
                     (7, A.FHASH_CANON_RAW, 500), (8, A.FHASH_VALUE_MASKED, 2)):
        p = A.SketchParams(A.ALGO_PROB3A, A.KMER32BIT, k, m, A.SIG_U32, A.HASHER_NOHASH, fh, 0, 0, 0, 0, 0)
        want = oracle.sketch(bases, off, p)
        ctx.profile_reset()
        ctx.profile_enable(True)
        got = np.asarray(ctx.sketch(bases, off, p))
        ctx.profile_enable(False)
        prof = ctx.profile_get()
        assert "k_sketch_smallk" in prof and ("k_pmh_points" in prof) == (split == "1")
        assert np.array_equal(got, want), (k, fh, m)
        pp = A.SketchParams.from_buffer_copy(p)
        pp.input_kind = A.INPUT_PACKED2
        assert np.array_equal(np.asarray(ctx.sketch(packed, off, pp, packed_offsets=poff)), want)
    monkeypatch.setenv("KMU_PMH_SMALLK", "0")
    assert np.array_equal(np.asarray(ctx.sketch(bases, off, p)), want)


def test_count_single_pass_overflow_on_a_fresh_context(oracle, monkeypatch):
    """The fall-back of the single-pass partition on a context whose scratch buffers do not exist yet: the attempt
    allocates "cnt.partA" / "cnt.partB" in its own sizes, the exact levels that take over allocate them anew (a route that
    kept pointers across the other's allocations wrote into freed memory: found by forcing the route on the whole suite).
    Segments of 60 % of their expected fill: far more items overflow than the spill list takes."""
    from kmerutils_amd import lib
    monkeypatch.setenv("KMU_COUNT_SEG", "2")
    monkeypatch.setenv("KMU_COUNT_SEG_PCT", "60")
    monkeypatch.setenv("KMU_COUNT_PATH", "partitioned")
    bases, off = synth.ont_reads(2500, 3_000_000, 0xC8)
    o = oracle.Counter(A.KMER64BIT, 31, 8, 1 << 22)
    o.add_reads(bases, off)
    fresh = lib.Context(0)
    try:
        c = fresh.counter(A.KMER64BIT, 31, 8, 12_000_000)  # 2^25 slots: two partition levels
        fresh.profile_reset()
        fresh.profile_enable(True)
        c.add_reads(bases, off)
        fresh.profile_enable(False)
        assert "k_part_hist1" in fresh.profile_get()  # the exact route ran
        wk, wc = o.dump(1)
        gk, gc = c.dump(1)
        assert np.array_equal(gk, wk) and np.array_equal(gc, wc)
        c.add_reads(bases, off)
        assert c.nb_distinct() == o.nb_distinct()
        c.close()
    finally:
        fresh.close()


def test_count_single_pass_spill_list(ctx, oracle, monkeypatch):
    """K-mers that occur thousands of times (homopolymer runs, a tandem repeat, the same reads three times over) overflow
    their segments of the single-pass partition: the items go to the spill list and are inserted into the finished table,
    the batch is NOT redone by the exact levels.  Counts equal to the oracle's, also for a second batch into the table."""
    monkeypatch.setenv("KMU_COUNT_SEG", "2")
    monkeypatch.setenv("KMU_COUNT_PATH", "partitioned")
    bases, off = synth.ont_reads(1200, 1_500_000, 0xC9)
    extra, eoff = oracle.concat([b"A" * 9000, b"ACGT" * 2000, b"T" * 7000, b"ACGGT" * 1500])
    parts, offs = [bases, bases, bases, extra], [off, off[1:] + off[-1], off[1:] + 2 * off[-1], eoff[1:] + 3 * off[-1]]
    bases = np.concatenate(parts)
    off = np.concatenate(offs)
    o = oracle.Counter(A.KMER64BIT, 31, 16, 1 << 22)
    o.add_reads(bases, off)
    c = ctx.counter(A.KMER64BIT, 31, 16, 12_000_000)
    ctx.profile_reset()
    ctx.profile_enable(True)
    c.add_reads(bases, off)
    ctx.profile_enable(False)
    prof = ctx.profile_get()
    assert "k_part_hist1" not in prof and "k_count_add_spill" in prof
    wk, wc = o.dump(1)
    gk, gc = c.dump(1)
    assert np.array_equal(gk, wk) and np.array_equal(gc, wc)
    c.add_reads(bases, off)
    o.add_reads(bases, off)
    wk, wc = o.dump(1)
    gk, gc = c.dump(1)
    assert np.array_equal(gk, wk) and np.array_equal(gc, wc)
    c.close()


@pytest.mark.parametrize("pct", ["100", "60"])
def test_count_single_pass_partition(ctx, oracle, monkeypatch, pct):
    """KMU_COUNT_SEG=2: the partitioned build with fixed-size segments and no histogram passes (what big batches take by
    themselves), forced on a batch the oracle can count; pct 60: segments smaller than their expected fill overflow, the
    flag is read before the table is touched and the exact route takes over.  Same table either way, also for a second
    batch into the same counter and with a non-ACGT byte in the reads."""
    import torch
    from kmerutils_amd.lib import KmuError
    monkeypatch.setenv("KMU_COUNT_SEG", "2")
    monkeypatch.setenv("KMU_COUNT_SEG_PCT", pct)
    monkeypatch.setenv("KMU_COUNT_PATH", "partitioned")
    dev = torch.device("cuda", 0)
    bases, off, lens = synth.ont_reads_device(1500, 9_000_000, 2_000_000, 0xC7, dev)
    hb, ho = bases[:int(off[-1])].cpu().numpy(), off.cpu().numpy().astype(np.uint64)
    nk = int(np.maximum(lens - 30, 0).sum())
    c = ctx.counter(A.KMER64BIT, 31, 8, 16_000_000)  # 2^25 slots: 8 192 regions, two partition levels
    ctx.profile_reset()
    ctx.profile_enable(True)
    c.add_reads(bases, off)
    ctx.profile_enable(False)
    prof = ctx.profile_get()
    assert ("k_part_hist1" in prof) == (pct == "60")  # the histogram passes only run on the exact route
    o = oracle.Counter(A.KMER64BIT, 31, 8, 1 << 22)
    o.add_reads(hb, ho)
    assert (c.nb_distinct(), c.nb_unique(), c.nb_occurrences()) == (o.nb_distinct(), o.nb_unique(), nk)
    wk, wc = o.dump(2)
    gk, gc = c.dump(2)
    assert np.array_equal(gk, wk) and np.array_equal(gc, wc)
    c.add_reads(bases, off)  # onto a table that holds something
    assert c.nb_occurrences() == 2 * nk and c.nb_distinct() == o.nb_distinct()
    c.close()
    bad = bases.clone()
    bad[12345] = ord("N")
    c = ctx.counter(A.KMER64BIT, 31, 8, 16_000_000)
    with pytest.raises(KmuError) as ei:
        c.add_reads(bad, off)
    assert ei.value.code == A.E_NON_ACGT
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k,kt,sig", [(21, A.KMER64BIT, A.SIG_U64), (31, A.KMER64BIT, A.SIG_U64), (12, A.KMER32BIT, A.SIG_U32)])
def test_short_reads_wave_per_read_multiset(ctx, oracle, monkeypatch, k, kt, sig):
    """A batch whose longest read has at most 256 k-mers takes k_multiset_short (one wave per read, a 512-slot table per wave)
    in front of k_pmh_points: 100-270 bp reads incl. homopolymers, tandem repeats, a read of exactly 256 k-mers, reads
    shorter than k; signatures equal to the oracle's, and to the general route's (KMU_PMH_SHORT=0)."""
    rng = np.random.default_rng(0x5107 + k)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(n))) for n in rng.integers(100, 256 + k - 1, size=3000)]
    seqs += [b"A" * 200, b"AC" * 120, b"ACGTT" * 50, b"T" * (255 + k), b"ACG"[:2] * 5, b"G" * (k - 1), b"C" * k]
    seqs.append(bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=255 + k)))  # exactly 256 k-mers
    bases, off = oracle.concat(seqs)
    p = A.SketchParams(A.ALGO_PROB3A, kt, k, 64, sig, 0, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    want = oracle.sketch(bases, off, p)
    monkeypatch.setenv("KMU_PMH_SPLIT", "1")
    ctx.profile_reset()
    ctx.profile_enable(True)
    got = ctx.sketch(bases, off, p)
    ctx.profile_enable(False)
    assert "k_multiset_short" in ctx.profile_get() and "k_pmh_points_short" in ctx.profile_get()
    assert np.array_equal(got, want)
    # a base outside ACGT in a short read: the reference panics, here the call fails
    from kmerutils_amd.lib import KmuError
    badb = bases.copy()
    badb[int(off[7]) + 33] = ord("N")
    with pytest.raises(KmuError) as ei:
        ctx.sketch(badb, off, p)
    assert ei.value.code == A.E_NON_ACGT
    monkeypatch.setenv("KMU_PMH_SHORT", "0")
    ctx.profile_reset()
    ctx.profile_enable(True)
    got2 = ctx.sketch(bases, off, p)
    ctx.profile_enable(False)
    assert "k_multiset_short" not in ctx.profile_get()
    assert np.array_equal(got2, want)
