"""Seeded sweep over the parameter space of kmu_sketch / kmu_count (small inputs): every configuration is compared
bit for bit with the oracle.  Complements the hand-picked cases of test_gpu_parity.py with odd sizes: sketch sizes 1 ..
4000, every k the value types allow, reads shorter than k, reads longer than one LDS pass, highly repetitive reads,
blocks, one-signature-for-all, both hashers, flags."""
import os

import numpy as np
import pytest

from kmerutils_amd import _abi as A

pytestmark = pytest.mark.gpu
# KMU_FUZZ_SCALE=10 runs ten times as many seeds per sweep (an occasional long run; the default keeps the suite short)
SCALE = int(os.environ.get("KMU_FUZZ_SCALE", "1"))


@pytest.fixture(scope="module")
def ctx():
    from kmerutils_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


def _reads(rng, n, kind):
    out = []
    for i in range(n):
        if kind == "short":
            L = int(rng.integers(1, 60))
        elif kind == "long":
            L = int(rng.integers(1, 40)) if i % 3 else int(rng.integers(20_000, 60_000))
        else:
            L = int(rng.integers(50, 4000))
        if kind == "repeat" and i % 2 == 0:
            unit = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(1, 9))).tobytes()
            s = (unit * (L // len(unit) + 1))[:L]
        else:
            s = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), size=L).tobytes()
        out.append(s)
    return out


def _kmer_choice(rng):
    t = int(rng.choice([A.KMER32BIT, A.KMER16B32BIT, A.KMER64BIT]))
    if t == A.KMER32BIT:
        return t, int(rng.integers(1, 15))
    if t == A.KMER16B32BIT:
        return t, 16
    return t, int(rng.integers(15, 32))


@pytest.mark.parametrize("seed", range(64 * SCALE))
def test_sketch_sweep(ctx, oracle, seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    kmer_type, k = _kmer_choice(rng)
    w32 = kmer_type != A.KMER64BIT
    algo = int(rng.choice([A.ALGO_PROB3A, A.ALGO_PROB3A, A.ALGO_SUPER, A.ALGO_BOTTOMK]))
    m = int(rng.choice([1, 2, 7, 64, 200, 333, 1000, 4000]))
    if algo == A.ALGO_SUPER:
        m = min(m, 1000)
    fhash = int(rng.choice([A.FHASH_IDENTITY_RAW, A.FHASH_VALUE_MASKED, A.FHASH_CANON_RAW, A.FHASH_CANON_INVHASH,
                            A.FHASH_INVHASH_RAW, A.FHASH_CANON_VALUE, A.FHASH_CANON_NTHASH]))
    if fhash == A.FHASH_CANON_NTHASH and w32:
        fhash = A.FHASH_CANON_INVHASH  # ntHash is a 64-bit value
    kind = str(rng.choice(["normal", "short", "long", "repeat"]))
    seqs = _reads(rng, int(rng.integers(1, 40)), kind)
    bases, off = oracle.concat(seqs)
    if algo == A.ALGO_PROB3A:
        sig, hasher = (A.SIG_U32 if w32 else A.SIG_U64), A.HASHER_NOHASH
    elif algo == A.ALGO_SUPER:
        sig, hasher = int(rng.choice([A.SIG_F32, A.SIG_F64])), int(rng.choice([A.HASHER_NOHASH, A.HASHER_FNV1A]))
    else:
        sig, hasher = A.SIG_U64, int(rng.choice([A.HASHER_NOHASH, A.HASHER_INT64HASH]))
    block = int(rng.choice([0, 0, 500, 2000])) if algo == A.ALGO_PROB3A and kind != "short" else 0
    mode = A.MODE_ALL_SEQS if (algo != A.ALGO_BOTTOMK and block == 0 and rng.random() < 0.2) else A.MODE_PER_SEQ
    flags = A.FLAG_RAND08 if rng.random() < 0.25 else 0
    p = A.SketchParams(algo, kmer_type, k, m, sig, hasher, fhash, block, mode, A.INPUT_ASCII, A.MEM_HOST, flags)
    want_counts = algo == A.ALGO_BOTTOMK
    try:
        want = oracle.sketch(bases, off, p, want_counts=want_counts)
    except oracle.OracleError as e:  # e.g. sketch_size 1: the device must refuse with the same status
        from kmerutils_amd.lib import KmuError
        with pytest.raises(KmuError) as g:
            ctx.sketch(bases, off, p, want_counts=want_counts)
        assert A.STATUS_NAMES[g.value.code] == str(e)
        return
    if block:
        bro = ctx.block_layout(np.ascontiguousarray(off, np.uint64), block)
        got = ctx.sketch(bases, off, p, block_row_offsets=bro)
    else:
        got = ctx.sketch(bases, off, p, want_counts=want_counts)
    if want_counts:
        assert np.array_equal(np.asarray(got[0]), want[0]) and np.array_equal(np.asarray(got[1]), want[1])
    else:
        a, b = np.asarray(got), np.asarray(want)
        assert a.shape == b.shape and a.tobytes() == b.tobytes(), (kmer_type, k, algo, m, fhash, kind, block, mode)
    if algo == A.ALGO_PROB3A and block == 0 and mode == A.MODE_PER_SEQ:
        # the route big batches take by default (multiset kernel -> (key, weight) lists -> k_pmh_points), forced
        monkeypatch.setenv("KMU_PMH_SPLIT", "1")
        a = np.asarray(ctx.sketch(bases, off, p))
        assert a.tobytes() == np.asarray(want).tobytes(), ("split", kmer_type, k, m, fhash, kind)


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_count_sweep(ctx, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    kmer_type, k = _kmer_choice(rng)
    bits = int(rng.choice([8, 16]))
    kind = str(rng.choice(["normal", "short", "long", "repeat"]))
    seqs = _reads(rng, int(rng.integers(1, 60)), kind)
    bases, off = oracle.concat(seqs)
    cap = max(1024, int(off[-1]))
    c = ctx.counter(kmer_type, k, bits, cap)
    o = oracle.Counter(kmer_type, k, bits, cap)
    for _ in range(int(rng.integers(1, 3))):  # the same reads twice: multiplicities double, saturation is exercised
        c.add_reads(bases, off)
        o.add_reads(bases, off)
    assert c.nb_distinct() == o.nb_distinct() and c.nb_unique() == o.nb_unique()
    gk, gc = c.dump(1)
    wk, wc = o.dump(1)
    assert np.array_equal(gk, wk) and np.array_equal(gc, wc)
    probe = np.concatenate([wk[:50], rng.integers(0, 1 << 20, size=20).astype(np.uint64)])
    assert np.array_equal(c.query(probe), o.query(probe))


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_aa_sketch_sweep(ctx, oracle, seed):
    rng = np.random.default_rng(3000 + seed)
    kmer_type = int(rng.choice([A.KMERAA32BIT, A.KMERAA64BIT]))
    k = int(rng.integers(1, 7 if kmer_type == A.KMERAA32BIT else 13))
    w32 = kmer_type == A.KMERAA32BIT
    algo = int(rng.choice([A.ALGO_PROB3A, A.ALGO_SUPER, A.ALGO_BOTTOMK]))
    m = int(rng.choice([2, 7, 128, 400, 800]))
    fhash = int(rng.choice([A.FHASH_IDENTITY_RAW, A.FHASH_VALUE_MASKED, A.FHASH_INVHASH_RAW]))
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    seqs = [rng.choice(aa, size=int(rng.integers(1, 30) if i % 5 == 0 else rng.integers(20, 1500))).tobytes()
            for i in range(int(rng.integers(1, 30)))]
    bases, off = oracle.concat(seqs)
    if algo == A.ALGO_PROB3A:
        sig, hasher = (A.SIG_U32 if w32 else A.SIG_U64), A.HASHER_NOHASH
    elif algo == A.ALGO_SUPER:
        sig, hasher = int(rng.choice([A.SIG_F32, A.SIG_F64])), int(rng.choice([A.HASHER_NOHASH, A.HASHER_FNV1A]))
    else:
        sig, hasher = A.SIG_U64, A.HASHER_NOHASH
    mode = A.MODE_ALL_SEQS if (algo != A.ALGO_BOTTOMK and rng.random() < 0.25) else A.MODE_PER_SEQ
    p = A.SketchParams(algo, kmer_type, k, m, sig, hasher, fhash, 0, mode, A.INPUT_ASCII, A.MEM_HOST, 0)
    want_counts = algo == A.ALGO_BOTTOMK
    try:
        want = oracle.sketch(bases, off, p, want_counts=want_counts)
    except oracle.OracleError as e:
        from kmerutils_amd.lib import KmuError
        with pytest.raises(KmuError) as g:
            ctx.sketch(bases, off, p, want_counts=want_counts)
        assert A.STATUS_NAMES[g.value.code] == str(e)
        return
    got = ctx.sketch(bases, off, p, want_counts=want_counts)
    if want_counts:
        assert np.array_equal(np.asarray(got[0]), want[0]) and np.array_equal(np.asarray(got[1]), want[1])
    else:
        assert np.asarray(got).tobytes() == np.asarray(want).tobytes(), (kmer_type, k, algo, m, fhash, mode)


@pytest.mark.parametrize("seed", range(32 * SCALE))
def test_dens_sketch_sweep(ctx, oracle, seed):
    """OptDens / RevOptDens: DNA and amino acids, f32 / f64, both hashers and index draws, per sequence and for all; sketch
    sizes from 2 to 6000 against reads from 1 base to 60 k (empty bins: from none to almost all)"""
    rng = np.random.default_rng(4000 + seed)
    algo = int(rng.choice([A.ALGO_OPTDENS, A.ALGO_REVOPTDENS]))
    m = int(rng.choice([1, 2, 7, 64, 200, 333, 1000, 6000]))
    sig = int(rng.choice([A.SIG_F32, A.SIG_F64]))
    hasher = int(rng.choice([A.HASHER_NOHASH, A.HASHER_FNV1A]))
    flags = A.FLAG_RAND08 if rng.random() < 0.25 else 0
    mode = A.MODE_ALL_SEQS if rng.random() < 0.3 else A.MODE_PER_SEQ
    if rng.random() < 0.25:
        kmer_type = int(rng.choice([A.KMERAA32BIT, A.KMERAA64BIT]))
        k = int(rng.integers(1, 7 if kmer_type == A.KMERAA32BIT else 13))
        fhash = int(rng.choice([A.FHASH_IDENTITY_RAW, A.FHASH_VALUE_MASKED, A.FHASH_INVHASH_RAW]))
        aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
        seqs = [rng.choice(aa, size=int(rng.integers(1, 30) if i % 5 == 0 else rng.integers(20, 1500))).tobytes()
                for i in range(int(rng.integers(1, 30)))]
    else:
        kmer_type, k = _kmer_choice(rng)
        fhash = int(rng.choice([A.FHASH_IDENTITY_RAW, A.FHASH_VALUE_MASKED, A.FHASH_CANON_RAW, A.FHASH_CANON_INVHASH,
                                A.FHASH_INVHASH_RAW, A.FHASH_CANON_VALUE]))
        seqs = _reads(rng, int(rng.integers(1, 40)), str(rng.choice(["normal", "short", "long", "repeat"])))
    bases, off = oracle.concat(seqs)
    p = A.SketchParams(algo, kmer_type, k, m, sig, hasher, fhash, 0, mode, A.INPUT_ASCII, A.MEM_HOST, flags)
    try:
        want = oracle.sketch(bases, off, p)
    except oracle.OracleError as e:
        from kmerutils_amd.lib import KmuError
        with pytest.raises(KmuError) as g:
            ctx.sketch(bases, off, p)
        assert A.STATUS_NAMES[g.value.code] == str(e)
        return
    got = np.asarray(ctx.sketch(bases, off, p))
    assert got.shape == want.shape and got.tobytes() == np.asarray(want).tobytes(), (algo, kmer_type, k, m, sig, fhash, mode)


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_counter_usage_sweep(ctx, oracle, seed, monkeypatch):
    """random sequences of counter operations against the oracle counter: batches of reads from the host, from the device
    (whole arrays and ranges of them), packed; explicit k-mer lists; merges of exported entries; the build path forced
    both ways; queries, totals and dumps in between; reset and reuse"""
    import torch
    rng = np.random.default_rng(5000 + seed)
    kmer_type, k = _kmer_choice(rng)
    bits = int(rng.choice([8, 16]))
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=30_000).tobytes()

    def batch():
        n = int(rng.integers(1, 120))
        seqs = []
        for _ in range(n):
            L = int(rng.integers(1, 900))
            s0 = int(rng.integers(0, len(genome) - L))
            seqs.append(genome[s0:s0 + L])   # overlapping samples of one genome: real multiplicities
        return oracle.concat(seqs)

    c = ctx.counter(kmer_type, k, bits, 1 << 17)
    o = oracle.Counter(kmer_type, k, bits, 1 << 17)
    for step in range(int(rng.integers(3, 9))):
        op = str(rng.choice(["host", "device", "range", "packed", "kmers", "merge", "reset", "forced"]))
        bases, off = batch()
        if op == "reset" and step > 0:
            c.reset()
            o = oracle.Counter(kmer_type, k, bits, 1 << 17)
            continue
        if op == "forced":
            monkeypatch.setenv("KMU_COUNT_PATH", str(rng.choice(["direct", "partitioned"])))
            c.add_reads(bases, off)
            monkeypatch.delenv("KMU_COUNT_PATH")
            o.add_reads(bases, off)
        elif op == "device":
            c.add_reads(torch.from_numpy(bases.copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda())
            o.add_reads(bases, off)
        elif op == "range":
            n = len(off) - 1
            a = int(rng.integers(0, n))
            b = int(rng.integers(a + 1, n + 1))
            c.add_reads(torch.from_numpy(bases.copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()[a:b + 1])
            sb = bases[int(off[a]):int(off[b])].copy()
            if sb.size == 0:
                sb = np.zeros(16, np.uint8)
            o.add_reads(sb, (off[a:b + 1] - off[a]).astype(np.uint64))
        elif op == "packed":
            packed, poff = ctx.pack2b(bases, off)
            c.add_reads(packed, off, A.INPUT_PACKED2, poff)
            o.add_reads(bases, off)
        elif op == "kmers":
            canon = oracle.kmer_hashes(bases, off, kmer_type, k, A.FHASH_CANON_VALUE)
            valid = [np.arange(int(off[i]), int(off[i + 1]) - k + 1) for i in range(len(off) - 1) if off[i + 1] - off[i] >= k]
            if valid:
                v = canon[np.concatenate(valid)]
                c.add_kmers(v.copy())
                o.add_kmers(v)
        elif op == "merge":
            c2 = ctx.counter(kmer_type, k, 16, 1 << 16)
            c2.add_reads(bases, off)
            kk, cc = c2.export_part(0, 1)
            c.merge_entries(kk.copy(), cc.copy())
            c2.close()
            o.add_reads(bases, off)
        else:
            c.add_reads(bases, off)
            o.add_reads(bases, off)
        if rng.random() < 0.5:
            probe = np.concatenate([o.dump(1)[0][:40], rng.integers(0, 1 << 30, size=10).astype(np.uint64)])
            assert np.array_equal(c.query(probe), o.query(probe)), (step, op)
    assert c.nb_distinct() == o.nb_distinct() and c.nb_unique() == o.nb_unique()
    for mc in (1, 2, 5):
        gk, gc = c.dump(mc)
        wk, wc = o.dump(mc)
        assert np.array_equal(gk, wk) and np.array_equal(gc, wc), mc
    c.close()
