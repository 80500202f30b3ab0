"""Pin the CPU oracle against every exact known-answer test the reference holds for the hot path
(SURVEY.md section 4 / 8c).  Vectors live in tests/golden/reference_kats.json with their reference citations."""
import json
import os

import numpy as np
import pytest

from kmerutils_amd import _abi as A

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
DNA = "ACGT"
AA = {1: "A", 2: "C", 3: "D", 4: "E", 5: "F", 6: "G", 7: "H", 8: "I", 9: "K", 10: "L", 11: "M", 12: "N", 13: "P",
      15: "Q", 16: "R", 17: "S", 18: "T", 19: "V", 20: "W", 21: "Y"}


def decode_dna(val, k):
    return "".join(DNA[(val >> (2 * (k - 1 - i))) & 3] for i in range(k))


def decode_aa(val, k):
    return "".join(AA[(val >> (5 * (k - 1 - i))) & 31] for i in range(k))


def from_str(O, kmer_type, s):
    """Kmer32bit::from_str / Kmer64bit::from_str: new(k) then push every base"""
    k = len(s)
    raw = O.lib().kmo_kmer_build(kmer_type, 0, k)
    for c in s.encode():
        raw = O.lib().kmo_kmer_push(kmer_type, raw, k, O.lib().kmo_encode2b(c))
    return raw


def test_alphabet_and_packing(oracle):
    L = oracle.lib()
    assert L.kmo_encode2b(ord("G")) == KAT["alphabet2b"]["G"]
    assert [L.kmo_encode2b(ord(c)) for c in "ACGTacgt"] == [0, 1, 2, 3, 0, 1, 2, 3]
    assert L.kmo_encode2b(ord("N")) == -1
    for v in KAT["pack2b"]:
        packed = oracle.pack2b(v["raw"].encode())
        assert len(packed) == v["bytes"]
        assert packed[0] == v["byte0"]  # ACGT -> 0x1B
        if "byte1_top2" in v:
            assert (packed[1] >> 6) & 3 == v["byte1_top2"]
            assert packed[1] & 0x3F == 0  # tail padded with 'A' = 0 (sequence.rs:66-71)
        assert len(v["raw"]) % 4 == v["last"]
        for i, c in enumerate(v["raw"]):
            assert L.kmo_get_base(packed.ctypes.data, i) == L.kmo_encode2b(ord(c))
    with pytest.raises(oracle.OracleError):
        oracle.pack2b(b"ACNT")
    assert L.kmo_count_non_acgt(np.frombuffer(b"ACNTxacgt", np.uint8).ctypes.data, 9) == 2


def test_encode_and_add_filters(oracle):
    L = oracle.lib()
    for v in KAT["encode_and_add"]:
        raw = np.frombuffer(v["raw"].encode(), np.uint8)
        out = np.zeros(16, np.uint8)
        kept = L.kmo_pack2b_filtered(raw.ctypes.data, raw.size, out.ctypes.data)
        s = "".join(DNA[L.kmo_get_base(out.ctypes.data, i)] for i in range(kept))
        assert s == v["kept"]


def test_revcomp_bit_patterns(oracle):
    L = oracle.lib()
    for a, b in zip(KAT["revcomp16"]["in"], KAT["revcomp16"]["out"]):
        assert L.kmo_kmer_revcomp(A.KMER16B32BIT, int(a, 2), 16) == int(b, 2)
    for v in KAT["revcomp_str"]:
        for t in (A.KMER32BIT, A.KMER64BIT):
            k = len(v["in"])
            rc = L.kmo_kmer_revcomp(t, from_str(oracle, t, v["in"]), k)
            assert rc == from_str(oracle, t, v["out"])
    o = KAT["ord32"]
    a, b = from_str(oracle, A.KMER32BIT, o["a"]), from_str(oracle, A.KMER32BIT, o["b"])
    assert bool(L.kmo_kmer_less(A.KMER32BIT, b, a)) == o["a_gt_b"]
    assert not L.kmo_kmer_less(A.KMER32BIT, a, a)


@pytest.mark.parametrize("name,kmer_type,k", [("seq80", A.KMER16B32BIT, 16), ("seq50", A.KMER16B32BIT, 16),
                                              ("seq50", A.KMER32BIT, 11), ("seq50", A.KMER64BIT, 21),
                                              ("seq48", A.KMER64BIT, 21), ("seq48", A.KMER16B32BIT, 16)])
def test_kmer_iteration_decodes_to_substrings(oracle, name, kmer_type, k):
    """kmergenerator.rs:596-732, 897-972: every k-mer decodes to seq[i..i+k]; the iterator stops after L-k+1"""
    s = KAT[name]
    bases, off = oracle.concat([s.encode()])
    out = oracle.kmer_hashes(bases, off, kmer_type, k, A.FHASH_IDENTITY_RAW)
    n = len(s) - k + 1
    for i in range(n):
        raw = int(out[i])
        if kmer_type == A.KMER32BIT:
            assert raw >> 28 == k  # k lives in the top nibble (kmer32bit.rs:212-216)
            raw &= 0x0FFFFFFF
        assert decode_dna(raw, k) == s[i:i + k]
    assert not out[n:].any()
    # same through the packed Sequence::new(raw,2) input
    packed = oracle.pack2b(s.encode())
    out2 = oracle.kmer_hashes(packed, off, kmer_type, k, A.FHASH_IDENTITY_RAW, A.INPUT_PACKED2,
                              np.zeros(2, np.uint64))
    assert (out2 == out).all()


def test_range_iterator(oracle):
    v = KAT["range_iter"]
    s = KAT[v["seq"]][v["begin"]:v["end"]]
    bases, off = oracle.concat([s.encode()])
    out = oracle.kmer_hashes(bases, off, A.KMER16B32BIT, v["k"], A.FHASH_IDENTITY_RAW)
    n = len(s) - v["k"] + 1
    assert n == v["n_kmers"]
    for i in range(n):
        assert decode_dna(int(out[i]), v["k"]) == KAT[v["seq"]][v["begin"] + i: v["begin"] + i + v["k"]]


def test_3mer_multiplicity_table(oracle):
    """kmergenerator.rs:777-850"""
    s = KAT["seq48"]
    bases, off = oracle.concat([s.encode()])
    out = oracle.kmer_hashes(bases, off, A.KMER32BIT, 3, A.FHASH_VALUE_MASKED)[:len(s) - 2]
    vals, cnt = np.unique(out, return_counts=True)
    got = {decode_dna(int(v), 3): int(c) for v, c in zip(vals, cnt)}
    assert got == KAT["kmer3_multiplicity"]["table"]


def test_weighted_kmer64(oracle):
    """kmergenerator.rs:853-894: k=15 on the repeated string, the first 9+... k-mers have weight 2"""
    s = KAT["seq72_repeat"]
    bases, off = oracle.concat([s.encode()])
    out = oracle.kmer_hashes(bases, off, A.KMER64BIT, 15, A.FHASH_IDENTITY_RAW)[:len(s) - 14]
    vals, cnt = np.unique(out, return_counts=True)
    for v, c in zip(vals, cnt):
        sub = decode_dna(int(v), 15)
        assert sum(1 for i in range(len(s) - 14) if s[i:i + 15] == sub) == c
    assert set(cnt.tolist()) == {1, 2}


def test_nthash_roll_equals_init(oracle):
    """nthash.rs:303-381: rolled hash == re-initialised hash at every position, forward and canonical"""
    L = oracle.lib()
    import ctypes as C
    s = KAT["seq80"].encode()
    k = 16
    buf = np.frombuffer(s, np.uint8)
    h = L.kmo_nthash_init_8b(buf.ctypes.data, k)
    fh, rh, st = C.c_uint64(), C.c_uint64(), C.c_uint8()
    hc = L.kmo_nthash_canonical_init_8b(buf.ctypes.data, k, C.byref(fh), C.byref(rh), C.byref(st))
    for i in range(1, len(s) - k):
        h = L.kmo_nthash_cycle_8b(h, k, s[i - 1], s[i - 1 + k])
        assert h == L.kmo_nthash_init_8b(buf[i:].ctypes.data, k)
        hc = L.kmo_nthash_canonical_cycle_8b(k, s[i - 1], s[i - 1 + k], C.byref(fh), C.byref(rh), C.byref(st))
        f2, r2, s2 = C.c_uint64(), C.c_uint64(), C.c_uint8()
        assert hc == L.kmo_nthash_canonical_init_8b(buf[i:].ctypes.data, k, C.byref(f2), C.byref(r2), C.byref(s2))
        assert (fh.value, rh.value, st.value) == (f2.value, r2.value, s2.value)
    # the per-position contract of KMU_FHASH_CANON_NTHASH_8B is exactly that
    bases, off = oracle.concat([s])
    out = oracle.kmer_hashes(bases, off, A.KMER16B32BIT, k, A.FHASH_CANON_NTHASH_8B)
    for i in range(len(s) - k + 1):
        assert out[i] == L.kmo_nthash_canonical_init_8b(buf[i:].ctypes.data, k, C.byref(fh), C.byref(rh), None)


def test_nthash_2bit_derived_kat(oracle):
    """SURVEY.md 8(a6) derived KAT for the proper seed table: first 16-mer of seq80"""
    L = oracle.lib()
    import ctypes as C
    s = KAT["seq80"][:16]
    val = 0
    for c in s:
        val = (val << 2) | DNA.index(c)
    fh, rh, st = C.c_uint64(), C.c_uint64(), C.c_uint8()
    h = L.kmo_nthash_canonical_2b(val, 16, C.byref(fh), C.byref(rh), C.byref(st))
    assert fh.value == 0x9840eab169670ddf and rh.value == 0x684a2ec1114d51c5
    assert h == rh.value and st.value == 1


def test_aa_kmers(oracle):
    aa = KAT["aa"]
    r = aa["range"]
    s = aa["seq149"][r["begin"]:r["end"]]
    for t in (A.KMERAA32BIT, A.KMERAA64BIT):
        bases, off = oracle.concat([s.encode()])
        out = oracle.kmer_hashes(bases, off, t, r["k"], A.FHASH_IDENTITY_RAW)
        got = [decode_aa(int(out[i]), r["k"]) for i in range(len(s) - r["k"] + 1)]
        assert got == r["kmers"]
    e = aa["end"]
    bases, off = oracle.concat([e["seq"].encode()])
    out = oracle.kmer_hashes(bases, off, A.KMERAA64BIT, e["k"], A.FHASH_IDENTITY_RAW)
    assert decode_aa(int(out[len(e["seq"]) - e["k"]]), e["k"]) == e["last"]
    # k = 12 works through the iterator (60-bit values)
    s12 = aa["seq149"]
    bases, off = oracle.concat([s12.encode()])
    out = oracle.kmer_hashes(bases, off, A.KMERAA64BIT, 12, A.FHASH_VALUE_MASKED)
    for i in (0, 7, len(s12) - 12):
        assert decode_aa(int(out[i]), 12) == s12[i:i + 12]
    with pytest.raises(oracle.OracleError):
        oracle.kmer_hashes(*oracle.concat([b"MTEQB"]), A.KMERAA32BIT, 3, A.FHASH_IDENTITY_RAW)  # 'B' panics


def test_counting_semantics(oracle):
    """kmercount.rs:1524-1559, 1580-1617"""
    s = KAT["seq80"]
    bases, off = oracle.concat([s.encode()])
    vk = oracle.kmer_hashes(bases, off, A.KMER16B32BIT, 16, A.FHASH_IDENTITY_RAW)[:65]
    rng = np.random.default_rng(7)
    c = oracle.Counter(A.KMER16B32BIT, 16, 8, 1024)
    c.add_kmers(vk[:2])
    c.add_kmers(vk[rng.integers(2, 65, 100000)])
    c.add_kmers(vk[1:2])
    exp = KAT["count"]
    assert c.query(vk[:2]).tolist() == [exp["kmer0_count"], exp["kmer1_count"]]
    c2 = oracle.Counter(A.KMER16B32BIT, 16, 8, 1024)
    c2.add_kmers(vk[rng.integers(32, 65, 100000)])
    got = c2.query(vk)
    assert (got[:32] == exp["never_inserted"]).all() and (got[32:] == exp["saturated"]).all()
    assert c2.nb_distinct() == len(set(vk[32:].tolist())) and c2.nb_unique() == 0
    c3 = oracle.Counter(A.KMER16B32BIT, 16, 16, 1024)
    c3.add_kmers(np.repeat(vk[:1], 70000))
    assert c3.query(vk[:1])[0] == 65535


def test_canonical_fhash_modes(oracle):
    L = oracle.lib()
    s = KAT["seq80"]
    rcs = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
    for t, k in ((A.KMER32BIT, 8), (A.KMER16B32BIT, 16), (A.KMER64BIT, 31)):
        b1, o1 = oracle.concat([s.encode()])
        b2, o2 = oracle.concat([rcs.encode()])
        n = len(s) - k + 1
        for fh in (A.FHASH_CANON_RAW, A.FHASH_CANON_INVHASH, A.FHASH_CANON_VALUE, A.FHASH_CANON_NTHASH):
            h1 = oracle.kmer_hashes(b1, o1, t, k, fh)[:n]
            h2 = oracle.kmer_hashes(b2, o2, t, k, fh)[:n]
            assert (h1 == h2[::-1]).all()  # strand invariance
        raw = oracle.kmer_hashes(b1, o1, t, k, A.FHASH_IDENTITY_RAW)[:n]
        ch = oracle.kmer_hashes(b1, o1, t, k, A.FHASH_CANON_INVHASH)[:n]
        for i in range(n):
            r = int(raw[i])
            rc = L.kmo_kmer_revcomp(t, r, k)
            canon = rc if L.kmo_kmer_less(t, rc, r) else r
            want = L.kmo_int32_hash(canon) if A.kmer_val_bytes(t) == 4 else L.kmo_int64_hash(canon)
            assert int(ch[i]) == want
    # hashers
    assert L.kmo_nohash_finish(0x01020304, 4) == 0x04030201
    assert L.kmo_nohash_finish(0x0102030405060708, 8) == 0x0807060504030201
    h = 0xcbf29ce484222325
    for byte in (0x04, 0x03, 0x02, 0x01):
        h = ((h ^ byte) * 0x100000001b3) & (2**64 - 1)
    assert L.kmo_fnv1a(0x01020304, 4) == h


def test_signature_comparison_oracle():
    """probminhash_get_jaccard_objects (seqsketchjaccard.rs:86-108) and minhash_distance (minhash.rs:134-190) on hand
    cases worked out from the reference's source."""
    from oracle import oracle as O
    a = np.array([1, 2, 3, 4, 5, 6], np.uint32)
    b = np.array([1, 9, 3, 9, 5, 9], np.uint32)
    assert O.sig_equal_count(a, b) == 3          # jp = 3 / 6
    assert O.sig_equal_count(a.astype(np.uint64), a.astype(np.uint64)) == 6
    # identical sketches: every step is a match, total stops at the sketch size
    s = np.array([2, 5, 7, 11], np.uint64)
    assert O.minhash_distance(s, s) == (4, 4, 4)
    # disjoint: the walk stops when `total` reaches len(sketch1); nothing in common
    assert O.minhash_distance(np.array([1, 2, 3, 4], np.uint64), np.array([10, 11, 12, 13], np.uint64)) == (0, 4, 4)
    # second sketch exhausted early: total is topped up from the unwalked part of sketch1, capped at its size
    assert O.minhash_distance(np.array([1, 2, 3, 4], np.uint64), np.array([1], np.uint64)) == (1, 4, 1)
    assert O.minhash_distance(np.array([], np.uint64), s) == (0, 0, 0)


def test_ingest_oracle_hand_cases():
    """readblockseq / parse_with_needletail rule (datasketcher.rs:364-371, io.rs:37-48): records with a non-ACGT byte are
    dropped and counted; the others keep their file order."""
    from oracle import oracle as O
    fq = (b"@r0\nACGTAC\n+\nIIIIII\n"
          b"@r1 has an N\nACNTAC\n+r1\nIIIIII\n"
          b"@r2 lower case is fine\nacgtTTGA\n+\n@@@@@@@@\n"      # quality line starting with '@'
          b"@r3\r\nGGGCCC\r\n+\r\nIIIIII\r\n"                      # CRLF
          b"@r4\nAC-T\n+\nIIII")                                   # no newline at the end; '-' is not ACGT
    bases, offs, info, idx = O.ingest_fastq(fq)
    assert info == dict(n_records=5, n_kept=3, kept_bases=20, n_bases=30, nb_bad_bases=2, nb_bad_reads=2)
    assert bytes(bases) == b"ACGTAC" + b"acgtTTGA" + b"GGGCCC"
    assert offs.tolist() == [0, 6, 14, 20] and idx.tolist() == [0, 2, 3]
    assert O.ingest_fastq(b"")[2]["n_records"] == 0
    for broken in (b"@r0\nACGT\n+\n", b"@r0\nACGT\n", b"r0\nACGT\n+\nIIII\n", b"@r0\nACGT\n-\nIIII\n",
                   b"@r0\nACGT\n+\nIIII\n\n"):
        with pytest.raises(O.OracleError):
            O.ingest_fastq(broken)


def test_ingest_fasta_oracle_hand_cases():
    """FASTA the way needletail hands it over (parse_fastx_file, io.rs:37): '>' at a line start opens a record,
    record.seq() is the following lines with the line ends removed; same drop rule as for FASTQ."""
    from oracle import oracle as O
    fa = (b">c0 first\nACGT\nAC\n"
          b">c1 with N\nACGT\nNNAC\n"
          b">c2 crlf\r\nGG\r\nCC\r\n"
          b">c3 header only\n"
          b">c4 > inside the header\n\nacgt\n\nT")          # empty lines, lower case, no newline at the end
    bases, offs, info, idx = O.ingest_fasta(fa)
    assert info == dict(n_records=5, n_kept=4, kept_bases=15, n_bases=23, nb_bad_bases=2, nb_bad_reads=1)
    assert bytes(bases) == b"ACGTAC" + b"GGCC" + b"" + b"acgtT"
    assert offs.tolist() == [0, 6, 10, 10, 15] and idx.tolist() == [0, 2, 3, 4]
    # the dispatcher goes by the first byte
    assert bytes(O.ingest_fastx(fa)[0]) == bytes(bases)
    assert bytes(O.ingest_fastx(b"@r\nACGT\n+\nIIII\n")[0]) == b"ACGT"
    for broken in (b"ACGT\n>c\nACGT\n", b"\n>c\nACGT\n", b"+\n"):
        with pytest.raises(O.OracleError):
            O.ingest_fastx(broken)
    with pytest.raises(O.OracleError):
        O.ingest_fasta(b"ACGT\n")


def test_ingest_fasta_oracle_against_line_by_line_restatement():
    """the oracle's FASTA reader against an independent few-line Python statement of the same rule, on seeded texts with
    every line width / CRLF / final-newline combination the GPU parity test uses"""
    from oracle import oracle as O
    from test_gpu_parity import _make_fasta
    for width, crlf, final_newline, n in ((60, False, True, 150), (1, False, True, 12), (80, True, True, 90),
                                          (0, False, False, 40), (17, True, False, 60), (70, False, True, 1)):
        fa = _make_fasta(np.random.default_rng(1000 + 7 * width + n), n, width, crlf, final_newline)
        records, cur = [], None
        for line in fa.replace(b"\r\n", b"\n").split(b"\n"):
            if line.startswith(b">"):
                cur = []
                records.append(cur)
            else:
                cur.append(line)
        seqs = [b"".join(r) for r in records]
        kept = [s for s in seqs if all(c in b"ACGTacgt" for c in s)]
        bases, offs, info, idx = O.ingest_fasta(fa)
        assert info["n_records"] == n and info["n_kept"] == len(kept) and info["n_bases"] == sum(map(len, seqs))
        assert bytes(bases) == b"".join(kept) and offs.tolist() == np.cumsum([0] + [len(s) for s in kept]).tolist()
        assert idx.tolist() == [i for i, s in enumerate(seqs) if all(c in b"ACGTacgt" for c in s)]


def test_seqminhash_reference_tests_on_the_oracle():
    """src/sketching/seqminhash.rs:127-258: sketches of two overlapping ranges [1, 65) and [35, 75) of the 80-base test
    string -- bottom-k (MinHashCount<u32, NoHashHasher> over int32_hash(canonical)) and SuperMinHash<f64, u32, NoHashHasher>.
    `resdist.3 == 20` (:190) is an exact assertion on minhash_distance; the others are thresholds."""
    from oracle import oracle as O
    S = b"TCAAAGGGAAACATTCAAAATCAGTATGCGCCCGTTCAGTTACGTATTGCTCTCGCTAATGAGATGGGCTGGGTACAGAG"
    bases, off = O.concat([S[1:65], S[35:75]])  # KmerSeqIterator::set_range(b, e): the k-mers inside [b, e)
    for kmer_type, k, total_check in ((A.KMER16B32BIT, 16, lambda t: t >= 3), (A.KMER32BIT, 10, lambda t: t == 20)):
        p = A.SketchParams(A.ALGO_BOTTOMK, kmer_type, k, 20, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
        h, c = O.sketch(bases, off, p, want_counts=True)
        common, total, i = O.minhash_distance(h[0], h[1])
        assert total_check(total) and common > 0                                   # :155, :190
    for kmer_type, k, m, thresh in ((A.KMER16B32BIT, 16, 50, 0.15), (A.KMER32BIT, 10, 20, 0.2)):
        p = A.SketchParams(A.ALGO_SUPER, kmer_type, k, m, A.SIG_F64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
        r = O.sketch(bases, off, p)
        assert float((r[0] == r[1]).mean()) >= thresh                              # :224, :256


# ---- the third-party layers under the sketch arithmetic, against their PUBLISHED vectors (tests/golden/reference_kats.json,
# "public_vectors"; Cargo.toml:74-89).  What stays unpinned after these is the control flow of probminhash 0.1 alone.
M64 = (1 << 64) - 1


def _inv_hash64shift(key, inv21, inv265):
    """the published inverse of Thomas Wang's hash64shift (probminhash::invhash::int64_invhash)"""
    tmp = (key - (key << 31)) & M64
    key = (key - (tmp << 31)) & M64
    tmp = key ^ (key >> 28)
    key = key ^ (tmp >> 28)
    key = (key * inv21) & M64
    tmp = key ^ (key >> 14)
    tmp = key ^ (tmp >> 14)
    tmp = key ^ (tmp >> 14)
    key = key ^ (tmp >> 14)
    key = (key * inv265) & M64
    tmp = key ^ (key >> 24)
    key = key ^ (tmp >> 24)
    tmp = ~key & M64
    tmp = ~(key - (tmp << 21)) & M64
    tmp = ~(key - (tmp << 21)) & M64
    return ~(key - (tmp << 21)) & M64


def _inv_hash32shift(y):
    """inverse of Thomas Wang's hash32shift, step by step (every step of the hash is a bijection of 32 bits)"""
    m = (1 << 32) - 1
    y ^= y >> 16
    y = (y * pow(2057, -1, 1 << 32)) & m
    x = y
    for _ in range(8):
        x = y ^ (x >> 4)
    y = (x * pow(5, -1, 1 << 32)) & m
    y = y ^ (y >> 12) ^ (y >> 24)
    return ((y + 1) * pow(32767, -1, 1 << 32)) & m


def test_public_vectors_xoshiro_splitmix(oracle):
    import ctypes as C
    L = oracle.lib()
    pv = KAT["public_vectors"]
    st = (C.c_uint64 * 4)(*pv["xoshiro256plusplus"]["state"])
    assert [L.kmo_xoshiro_next(st) for _ in range(10)] == [int(x) for x in pv["xoshiro256plusplus"]["outputs"]]
    for case in pv["splitmix64"]:
        # Xoshiro256PlusPlus::seed_from_u64 fills the state with the first four SplitMix64 outputs of the seed; the fifth is
        # the first output of the stream continued from where the fourth left it
        seed = int(case["seed"])
        L.kmo_xoshiro_seed(seed, st)
        assert list(st) == [int(x) for x in case["outputs"][:4]]
        L.kmo_xoshiro_seed((seed + 4 * 0x9E3779B97F4A7C15) & M64, st)
        assert st[0] == int(case["outputs"][4])


def test_public_vectors_fnv_and_wang_hashes(oracle):
    L = oracle.lib()
    pv = KAT["public_vectors"]
    for text, want in pv["fnv1a64"]["vectors"]:
        b = text.encode()
        assert L.kmo_fnv1a(int.from_bytes(b, "little"), len(b)) == int(want, 16), text
    hv = pv["hash64shift_inverse"]
    inv21, inv265 = int(hv["inv21"]), int(hv["inv265"])
    assert (21 * inv21) & M64 == 1 and (265 * inv265) & M64 == 1
    rng = np.random.default_rng(1)
    for x in [int(v) for v in hv["inputs"]] + [int(v) for v in rng.integers(0, 1 << 63, 2000, dtype=np.uint64)]:
        assert _inv_hash64shift(L.kmo_int64_hash(x), inv21, inv265) == x
    for x in [0, 1, 2, 255, 65535, 1 << 31, (1 << 32) - 1] + [int(v) for v in rng.integers(0, 1 << 32, 2000, dtype=np.uint64)]:
        assert _inv_hash32shift(L.kmo_int32_hash(x)) == x
    # hash64shift(0) by hand: ~0 + 0 = 2^64 - 1, then the six mixing steps of the published function
    k = M64
    k ^= k >> 24
    k = (k + (k << 3) + (k << 8)) & M64
    k ^= k >> 14
    k = (k + (k << 2) + (k << 4)) & M64
    k ^= k >> 28
    k = (k + (k << 31)) & M64
    assert L.kmo_int64_hash(0) == k


def test_public_vectors_uniform_f64(oracle):
    import ctypes as C
    L = oracle.lib()
    for word, want in KAT["public_vectors"]["uniform_f64"]["vectors"]:
        w = int(word)
        # a state whose next output is `w`: s0 = s1 = s2 = 0, rotl(s3, 23) = w
        st = (C.c_uint64 * 4)(0, 0, 0, ((w >> 23) | (w << 41)) & M64)
        assert L.kmo_unif01_f64(st) == want, word
