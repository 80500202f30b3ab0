"""GPU parity of the rest of the KmerGenerationPattern surface and of nthash.rs through the C-ABI: ranges
(generate_kmer_pattern_in_range), distributions (generate_kmer_distribution) and per-position ntHash with strand,
forward / rcomp modes and the multi-hash expansion.  The reference's own tests (kmergenerator.rs:661-700, :777-850,
:853-894; nthash.rs:303-381) run through the HIP path here; the oracle checks everything else bit for bit."""
import json
import os

import numpy as np
import pytest

from kmerutils_amd import _abi as A
from kmerutils_amd import synth

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
DNA = "ACGT"
RAGGED = [1, 2, 7, 15, 16, 17, 30, 31, 32, 33, 47, 48, 49, 63, 64, 65, 100, 150, 151, 991, 992, 993, 1000, 1007, 1008, 1009,
          1023, 1024, 1025, 1039, 1040, 1041, 1984, 1985, 2047, 2048, 2079, 5000, 12345]


def decode_dna(val, k):
    return "".join(DNA[(val >> (2 * (k - 1 - i))) & 3] for i in range(k))


@pytest.fixture(scope="module")
def ctx():
    from kmerutils_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


def ragged_dna(seed, lens):
    rng = np.random.default_rng(seed)
    return [bytes(synth.ACGT[rng.integers(0, 4, L)]) for L in lens]


def canon_pairs(kk, cc, do, n):
    """per sequence: (values, counts) sorted by value"""
    out = []
    for i in range(n):
        k = np.asarray(kk[int(do[i]):int(do[i + 1])]).astype(np.uint64)
        c = np.asarray(cc[int(do[i]):int(do[i + 1])]).astype(np.uint32)
        o = np.argsort(k, kind="stable")
        out.append((k[o], c[o]))
    return out


# ---- ranges ---------------------------------------------------------------------------------------------------------
def test_reference_range_test_through_hip(ctx):
    """kmergenerator.rs:661-700 on the device"""
    v = KAT["range_iter"]
    s = KAT[v["seq"]]
    bases = np.frombuffer(s.encode(), np.uint8).copy()
    off = np.array([0, len(s)], np.uint64)
    out = ctx.kmer_hashes_range(bases, off, A.KMER16B32BIT, v["k"], A.FHASH_IDENTITY_RAW, np.array([v["begin"]], np.uint64),
                                np.array([v["end"]], np.uint64))
    n = v["end"] - v["begin"] - v["k"] + 1
    assert n == v["n_kmers"]
    assert [decode_dna(int(out[v["begin"] + i]), v["k"]) for i in range(n)] == [s[v["begin"] + i: v["begin"] + i + v["k"]] for i in range(n)]
    assert not out[:v["begin"]].any() and not out[v["begin"] + n:].any()


@pytest.mark.parametrize("kmer_type,k,fh", [(A.KMER32BIT, 5, A.FHASH_CANON_INVHASH), (A.KMER16B32BIT, 16, A.FHASH_IDENTITY_RAW),
                                            (A.KMER64BIT, 31, A.FHASH_CANON_VALUE), (A.KMER64BIT, 21, A.FHASH_CANON_NTHASH)])
def test_ranges_vs_oracle(ctx, oracle, kmer_type, k, fh):
    import torch
    seqs = ragged_dna(40 + k, RAGGED)
    bases, off = oracle.concat(seqs)
    rng = np.random.default_rng(k)
    L = np.diff(off.astype(np.int64))
    for rnd in range(3):
        rb = np.array([rng.integers(0, l) for l in L], np.uint64)
        re = np.array([rng.integers(b + 1, l + 1) for b, l in zip(rb, L)], np.uint64)
        if rnd == 0:  # whole sequences == the plain call
            rb[:] = 0
            re[:] = L
        want = oracle.kmer_hashes_range(bases, off, kmer_type, k, fh, rb, re)
        got = ctx.kmer_hashes_range(bases, off, kmer_type, k, fh, rb, re)
        assert np.array_equal(got, want)
        if rnd == 0:
            assert np.array_equal(got, ctx.kmer_hashes(bases, off, kmer_type, k, fh))
        # device buffers, packed input
        d = lambda a, dt: torch.from_numpy(a.astype(dt)).cuda()
        got_d = ctx.kmer_hashes_range(d(bases, np.uint8), d(off, np.int64), kmer_type, k, fh, d(rb, np.int64), d(re, np.int64))
        assert np.array_equal(got_d.cpu().numpy().view(np.uint64), want)
        packed, poff = ctx.pack2b(bases, off)
        got_p = ctx.kmer_hashes_range(packed, off, kmer_type, k, fh, rb, re, A.INPUT_PACKED2, poff)
        assert np.array_equal(got_p, want)


def test_range_aa_and_errors(ctx, oracle):
    import torch
    from kmerutils_amd.lib import KmuError
    prot, poff = synth.protein_seqs(40, 7)
    L = np.diff(poff.astype(np.int64))
    rng = np.random.default_rng(3)
    rb = np.array([rng.integers(0, l) for l in L], np.uint64)
    re = np.array([rng.integers(b + 1, l + 1) for b, l in zip(rb, L)], np.uint64)
    for kt, k in ((A.KMERAA32BIT, 5), (A.KMERAA64BIT, 12)):
        want = oracle.kmer_hashes_range(prot, poff.astype(np.uint64), kt, k, A.FHASH_VALUE_MASKED, rb, re)
        got = ctx.kmer_hashes_range(prot, poff.astype(np.uint64), kt, k, A.FHASH_VALUE_MASKED, rb, re)
        assert np.array_equal(got, want)
    # IterSequence::set_range Err (sequence.rs:563-565): host and device buffers alike
    bases = np.frombuffer(KAT["seq50"].encode(), np.uint8).copy()
    off = np.array([0, 50], np.uint64)
    for b, e in ((5, 5), (9, 3), (0, 51), (50, 60)):
        with pytest.raises(KmuError) as ei:
            ctx.kmer_hashes_range(bases, off, A.KMER32BIT, 8, A.FHASH_IDENTITY_RAW, np.array([b], np.uint64), np.array([e], np.uint64))
        assert ei.value.code == A.E_BAD_ARG
        with pytest.raises(KmuError) as ei:
            ctx.kmer_hashes_range(torch.from_numpy(bases).cuda(), torch.tensor([0, 50]).cuda(), A.KMER32BIT, 8, A.FHASH_IDENTITY_RAW,
                                  torch.tensor([b]).cuda(), torch.tensor([e]).cuda())
        assert ei.value.code == A.E_BAD_ARG
    # a good call afterwards starts from a clean error word
    ctx.kmer_hashes_range(bases, off, A.KMER32BIT, 8, A.FHASH_IDENTITY_RAW, np.array([1], np.uint64), np.array([3], np.uint64))


# ---- distributions ----------------------------------------------------------------------------------------------------
def test_reference_distribution_tests_through_hip(ctx):
    """kmergenerator.rs:777-850 (31-entry 3-mer table) and :853-894 (weighted 15-mers) on the device"""
    s = KAT["seq48"]
    kk, cc, do = ctx.kmer_distribution(np.frombuffer(s.encode(), np.uint8).copy(), np.array([0, len(s)], np.uint64), A.KMER32BIT, 3,
                                       A.FHASH_VALUE_MASKED)
    assert {decode_dna(int(v), 3): int(c) for v, c in zip(kk, cc)} == KAT["kmer3_multiplicity"]["table"] and list(do) == [0, 31]
    s = KAT["seq72_repeat"]
    kk, cc, _ = ctx.kmer_distribution(np.frombuffer(s.encode(), np.uint8).copy(), np.array([0, len(s)], np.uint64), A.KMER64BIT, 15)
    assert int(cc.sum()) == len(s) - 14 and set(cc.tolist()) == {1, 2}
    for v, c in zip(kk, cc):
        sub = decode_dna(int(v), 15)
        assert sum(1 for i in range(len(s) - 14) if s[i:i + 15] == sub) == c


@pytest.mark.parametrize("kmer_type,k,fh", [(A.KMER32BIT, 4, A.FHASH_IDENTITY_RAW), (A.KMER32BIT, 8, A.FHASH_CANON_INVHASH),
                                            (A.KMER16B32BIT, 16, A.FHASH_IDENTITY_RAW), (A.KMER64BIT, 31, A.FHASH_CANON_VALUE)])
def test_distribution_vs_oracle(ctx, oracle, kmer_type, k, fh):
    import torch
    seqs = ragged_dna(60 + k, RAGGED)
    seqs += [b"A" * 3000, b"ACGT" * 2500, seqs[-1] * 3, b"ACG"]  # poly-A, tandem repeat, a read three times over, < k
    bases, off = oracle.concat(seqs)
    want = canon_pairs(*oracle.kmer_distribution(bases, off, kmer_type, k, fh), len(seqs))
    for dev in (False, True):
        if dev:
            kk, cc, do = ctx.kmer_distribution(torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), kmer_type, k, fh)
            kk, cc, do = kk.cpu().numpy().view(np.uint64), cc.cpu().numpy().view(np.uint32), do.cpu().numpy()
        else:
            kk, cc, do = ctx.kmer_distribution(bases, off, kmer_type, k, fh)
        got = canon_pairs(kk, cc, do, len(seqs))
        for (gk, gc), (wk, wc) in zip(got, want):
            assert np.array_equal(gk, wk) and np.array_equal(gc, wc)


def test_distribution_long_sequence_and_aa(ctx, oracle):
    """a sequence of many hash passes (300 k bases: ~290 work items) and amino-acid k-mers (kmeraa.rs:752-778)"""
    long = ragged_dna(0xD1, [300_000, 5000])
    long.append(long[1] + long[1])  # every k-mer of the second sequence at least twice
    b2, o2 = oracle.concat(long)
    want = canon_pairs(*oracle.kmer_distribution(b2, o2, A.KMER64BIT, 21, A.FHASH_CANON_INVHASH), len(o2) - 1)
    got = canon_pairs(*ctx.kmer_distribution(b2, o2, A.KMER64BIT, 21, A.FHASH_CANON_INVHASH), len(o2) - 1)
    for (gk, gc), (wk, wc) in zip(got, want):
        assert np.array_equal(gk, wk) and np.array_equal(gc, wc)
    prot, poff = synth.protein_seqs(50, 11)
    for kt, k in ((A.KMERAA32BIT, 3), (A.KMERAA64BIT, 7)):
        want = canon_pairs(*oracle.kmer_distribution(prot, poff.astype(np.uint64), kt, k, A.FHASH_IDENTITY_RAW), 50)
        got = canon_pairs(*ctx.kmer_distribution(prot, poff.astype(np.uint64), kt, k, A.FHASH_IDENTITY_RAW), 50)
        for (gk, gc), (wk, wc) in zip(got, want):
            assert np.array_equal(gk, wk) and np.array_equal(gc, wc)


# ---- ntHash -------------------------------------------------------------------------------------------------------------
def test_nthash_reference_property_through_hip(ctx, oracle):
    """nthash.rs:303-381 on the device: at every position of seq80 the value equals the reference's init function of the
    k-mer that starts there (= the rolled value), forward and canonical, strand included"""
    import ctypes as C
    L = oracle.lib()
    s = KAT["seq80"].encode()
    buf = np.frombuffer(s, np.uint8).copy()
    off = np.array([0, len(s)], np.uint64)
    for k in (5, 16, 31):
        can, strand = ctx.nthash(buf, off, k, table=A.NTHASH_TABLE_8B)
        fwd = ctx.nthash(buf, off, k, mode=A.NTHASH_FORWARD, table=A.NTHASH_TABLE_8B, want_strand=False)
        for i in range(len(s) - k + 1):
            f, r, st = C.c_uint64(), C.c_uint64(), C.c_uint8()
            h = L.kmo_nthash_canonical_init_8b(buf[i:].ctypes.data, k, C.byref(f), C.byref(r), C.byref(st))
            assert (int(can[i, 0]), int(strand[i]), int(fwd[i, 0])) == (h, st.value, L.kmo_nthash_init_8b(buf[i:].ctypes.data, k))
    # 2-bit table, derived KAT of SURVEY 8(a6)
    h, st = ctx.nthash(buf, off, 16)
    assert int(h[0, 0]) == 0x684a2ec1114d51c5 and int(st[0]) == 1


@pytest.mark.parametrize("k", [1, 5, 15, 16, 17, 21, 31, 32])
def test_nthash_vs_oracle(ctx, oracle, k):
    import torch
    seqs = ragged_dna(900 + k, RAGGED)
    bases, off = oracle.concat(seqs)
    packed, poff = ctx.pack2b(bases, off)
    for table in (A.NTHASH_TABLE_2B, A.NTHASH_TABLE_8B):
        for mode in (A.NTHASH_CANONICAL, A.NTHASH_FORWARD, A.NTHASH_RCOMP):
            nh = 1 if mode else 4
            wh, ws = oracle.nthash(bases, off, k, nh, mode, table)
            gh, gs = ctx.nthash(bases, off, k, nh, mode, table)
            assert np.array_equal(gh, wh) and np.array_equal(gs, ws), (table, mode)
            if table == A.NTHASH_TABLE_2B:
                ph, ps = ctx.nthash(packed, off, k, nh, mode, table, A.INPUT_PACKED2, poff)
                assert np.array_equal(ph, wh) and np.array_equal(ps, ws)
    dh, dsd = ctx.nthash(torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), k, 3)
    wh, ws = oracle.nthash(bases, off, k, 3)
    assert np.array_equal(dh.cpu().numpy().view(np.uint64), wh) and np.array_equal(dsd.cpu().numpy(), ws)
    # the canonical single value is what the sketch / hash closures KMU_FHASH_CANON_NTHASH compute
    if k <= 31:
        kt = A.KMER32BIT if k <= 14 else A.KMER16B32BIT if k == 16 else A.KMER64BIT
        assert np.array_equal(ctx.kmer_hashes(bases, off, kt, k, A.FHASH_CANON_NTHASH), wh[:, 0])


@pytest.mark.parametrize("pre", [37, 1024 + 16])
def test_nthash_of_reads_that_start_inside_the_device_buffer(ctx, oracle, pre):
    """offsets[0] > 0 (a range of a larger read set in HBM): the output is indexed by the absolute position, what lies before the
    first read is neither looked at (non-ACGT bytes there are no error) nor written -- the positions that start a k-mer come from
    the bit mask of flat_novalid, which marks everything before offsets[0]"""
    import torch
    seqs = ragged_dna(700, RAGGED)
    bases, off = oracle.concat(seqs)
    buf = np.concatenate([np.full(pre, ord("N"), np.uint8), bases])
    off2 = (off.astype(np.int64) + pre)
    for k, nh in ((21, 1), (31, 2)):
        wh, ws = oracle.nthash(bases, off, k, nh)
        gh, gs = ctx.nthash(torch.from_numpy(buf).cuda(), torch.from_numpy(off2).cuda(), k, nh)
        gh, gs = gh.cpu().numpy().view(np.uint64), gs.cpu().numpy()
        assert np.array_equal(gh[pre:], wh) and np.array_equal(gs[pre:], ws), (k, nh)
        assert not gh[:pre].any() and not gs[:pre].any()


def test_nthash_errors(ctx):
    from kmerutils_amd.lib import KmuError
    bases = np.frombuffer(b"ACGTNACGTACGT", np.uint8).copy()
    off = np.array([0, 13], np.uint64)
    with pytest.raises(KmuError) as ei:
        ctx.nthash(bases, off, 4)
    assert ei.value.code == A.E_NON_ACGT
    with pytest.raises(KmuError) as ei:
        ctx.nthash(bases, off, 33)
    assert ei.value.code == A.E_BAD_K


def test_device_wang_hashes_against_the_published_inverse(ctx, oracle):
    """probminhash::invhash::int32_hash / int64_hash on the device (kmu_kmer_hashes, FHASH_INVHASH_RAW / CANON_INVHASH): the
    published inverse of Thomas Wang's hash64shift (tests/golden/reference_kats.json, public_vectors; restated for 32 bits in
    tests/test_oracle_kat.py) takes every device value back to the k-mer it was made of, and the device values are the oracle's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kat", os.path.join(os.path.dirname(__file__), "test_oracle_kat.py"))
    kat = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kat)
    hv = kat.KAT["public_vectors"]["hash64shift_inverse"]
    inv21, inv265 = int(hv["inv21"]), int(hv["inv265"])
    bases, off = synth.ont_reads(4, 50_000, 0xA7)
    n = int(off[-1])
    for kmer_type, k in ((A.KMER64BIT, 31), (A.KMER64BIT, 17), (A.KMER32BIT, 12), (A.KMER16B32BIT, 16)):
        raw = ctx.kmer_hashes(bases, off, kmer_type, k, A.FHASH_IDENTITY_RAW)[:n]
        hashed = ctx.kmer_hashes(bases, off, kmer_type, k, A.FHASH_INVHASH_RAW)[:n]
        canon = ctx.kmer_hashes(bases, off, kmer_type, k, A.FHASH_CANON_RAW)[:n]
        chash = ctx.kmer_hashes(bases, off, kmer_type, k, A.FHASH_CANON_INVHASH)[:n]
        assert np.array_equal(hashed, oracle.kmer_hashes(bases, off, kmer_type, k, A.FHASH_INVHASH_RAW)[:n])
        valid = np.zeros(n, bool)
        for i in range(len(off) - 1):
            valid[int(off[i]):max(int(off[i]), int(off[i + 1]) - k + 1)] = True
        idx = np.flatnonzero(valid)[::37]
        w32 = A.kmer_val_bytes(kmer_type) == 4
        for i in idx:
            inv = kat._inv_hash32shift if w32 else (lambda y: kat._inv_hash64shift(y, inv21, inv265))
            assert inv(int(hashed[i])) == int(raw[i]) and inv(int(chash[i])) == int(canon[i]), (kmer_type, k, i)


def test_kmer_hashes_tail_positions_by_memory_kind(ctx, oracle):
    """include/kmu.h, kmu_kmer_hashes: the positions that start no k-mer (the last k - 1 of a sequence, every position of a
    sequence shorter than k) come back ZERO in a KMU_MEM_HOST array and are LEFT AS THEY WERE in a KMU_MEM_DEVICE array; the
    k-mer positions are the oracle's either way (pinned, so that a caller's sentinel means the same tomorrow)."""
    import torch
    k = 21
    seqs = ragged_dna(3, [100, 5, 21, 20, 64, 1, 300])
    bases, off = oracle.concat(seqs)
    want = oracle.kmer_hashes(bases, off, A.KMER64BIT, k, A.FHASH_CANON_INVHASH)
    n = int(off[-1])
    starts = np.zeros(n, bool)
    for i in range(len(off) - 1):
        starts[int(off[i]):max(int(off[i]), int(off[i + 1]) - k + 1)] = True
    sentinel = 0x5E5E5E5E5E5E5E5E
    host = np.full(n, sentinel, np.uint64)
    ctx.kmer_hashes(bases, off, A.KMER64BIT, k, A.FHASH_CANON_INVHASH, out=host)
    assert np.array_equal(host[starts], want[:n][starts]) and (host[~starts] == 0).all()
    dev = torch.full((n,), sentinel, dtype=torch.int64, device="cuda")
    ctx.kmer_hashes(torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), A.KMER64BIT, k, A.FHASH_CANON_INVHASH, out=dev)
    ctx.synchronize()
    got = dev.cpu().numpy().view(np.uint64)
    assert np.array_equal(got[starts], want[:n][starts]) and (got[~starts] == sentinel).all()


def test_every_byte_value_at_every_place_of_a_chunk(ctx, oracle, monkeypatch):
    """The kernels turn 16 ASCII bytes into a code word four bytes at a time (pack16_ascii, kmu_device.h): every byte value in
    each of the sixteen places of a chunk must give what the byte-wise rule of alphabet.rs:119-127 gives -- the code of
    ACGTacgt (the oracle's hashes), the reference's panic (KMU_E_NON_ACGT) for the 248 other values."""
    from kmerutils_amd.lib import KmuError
    rng = np.random.default_rng(5)
    letters = np.frombuffer(b"ACGTacgt", np.uint8)
    base = letters[rng.integers(0, 8, 96)].copy()
    off = np.array([0, 96], np.uint64)
    valid = set(b"ACGTacgt")
    refused = 0
    for place in range(16):
        for b in range(256):
            seq = base.copy()
            seq[32 + place] = b
            if b in valid:
                want = oracle.kmer_hashes(seq, off, A.KMER64BIT, 21, A.FHASH_CANON_INVHASH)
                assert np.array_equal(ctx.kmer_hashes(seq, off, A.KMER64BIT, 21, A.FHASH_CANON_INVHASH), want), (place, b)
            else:
                with pytest.raises(KmuError) as ei:
                    ctx.kmer_hashes(seq, off, A.KMER64BIT, 21, A.FHASH_CANON_INVHASH)
                assert ei.value.code == A.E_NON_ACGT, (place, b)
                refused += 1
    assert refused == 16 * 248
    # ... and through the count's level 1 (flat_step_words) and the sketch's staged words: one bad byte anywhere refuses the batch
    from kmerutils_amd import lib
    monkeypatch.setenv("KMU_COUNT_PATH", "partitioned")
    seqs = [bytes(synth.ACGT[rng.integers(0, 4, 3000)]) for _ in range(40)]
    bases, offs = oracle.concat(seqs)
    for pos in (0, 1, 15, 16, 17, 2999, 3000, 40 * 3000 - 1, 77777):
        for b in (ord("N"), 0, 0x40, 0x42, 0x55, 0x61 ^ 0x80, 0xFF):
            bad = bases.copy()
            bad[pos] = b
            c = lib.Counter(ctx, A.KMER64BIT, 31, capacity_hint=200000)
            with pytest.raises(KmuError) as ei:
                c.add_reads(bad, offs)
            assert ei.value.code == A.E_NON_ACGT, (pos, b)
            c.close()
