"""The oracle's restatements of the rest of the KmerGenerationPattern surface (ranges, distributions) and of nthash.rs
(strand, forward / rcomp, multi-hash) against the reference's own known-answer tests and invariants."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from kmerutils_amd import _abi as A

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
DNA = "ACGT"


def decode_dna(val, k):
    return "".join(DNA[(val >> (2 * (k - 1 - i))) & 3] for i in range(k))


def test_range_iterator_set_range(oracle):
    """kmergenerator.rs:661-700: k-mers of a range of seq50 through set_range itself (not a slice of the string)"""
    v = KAT["range_iter"]
    s = KAT[v["seq"]]
    bases, off = oracle.concat([s.encode()])
    out = oracle.kmer_hashes_range(bases, off, A.KMER16B32BIT, v["k"], A.FHASH_IDENTITY_RAW, [v["begin"]], [v["end"]])
    n = v["end"] - v["begin"] - v["k"] + 1
    assert n == v["n_kmers"]
    for i in range(n):
        assert decode_dna(int(out[v["begin"] + i]), v["k"]) == s[v["begin"] + i: v["begin"] + i + v["k"]]
    assert not out[:v["begin"]].any() and not out[v["begin"] + n:].any()
    # packed input walks the same bytes
    packed = oracle.pack2b(s.encode())
    out2 = oracle.kmer_hashes_range(packed, off, A.KMER16B32BIT, v["k"], A.FHASH_IDENTITY_RAW, [v["begin"]], [v["end"]],
                                    A.INPUT_PACKED2, np.zeros(2, np.uint64))
    assert np.array_equal(out, out2)


@pytest.mark.parametrize("kmer_type,k", [(A.KMER32BIT, 5), (A.KMER16B32BIT, 16), (A.KMER64BIT, 21)])
def test_range_equals_slice_every_range(oracle, kmer_type, k):
    """IterSequence::set_range byte/bit arithmetic (sequence.rs:562-585) on every (begin, end) of a 41-base sequence,
    incl. ends 1..3 (the release-build wrap of :574) and non-multiples of 4 on both sides"""
    s = KAT["seq50"][:41]
    bases, off = oracle.concat([s.encode()])
    full = oracle.kmer_hashes(bases, off, kmer_type, k, A.FHASH_CANON_INVHASH)
    for b in range(0, 41):
        for e in range(b + 1, 42):
            out = oracle.kmer_hashes_range(bases, off, kmer_type, k, A.FHASH_CANON_INVHASH, [b], [e])
            n = max(0, e - b - k + 1)
            assert np.array_equal(out[b:b + n], full[b:b + n]) and not out[:b].any() and not out[b + n:].any()


def test_range_errors(oracle):
    """set_range: Err if end <= begin or end > size (sequence.rs:563-565); the callers unwrap"""
    bases, off = oracle.concat([KAT["seq50"].encode()])
    for b, e in ((5, 5), (9, 3), (0, 51), (50, 60)):
        with pytest.raises(oracle.OracleError) as ei:
            oracle.kmer_hashes_range(bases, off, A.KMER32BIT, 8, A.FHASH_IDENTITY_RAW, [b], [e])
        assert ei.value.code == A.E_BAD_ARG


def test_distribution_3mer_table(oracle):
    """kmergenerator.rs:777-850: the 31-entry multiplicity table of seq48"""
    bases, off = oracle.concat([KAT["seq48"].encode()])
    kk, cc, do = oracle.kmer_distribution(bases, off, A.KMER32BIT, 3, A.FHASH_VALUE_MASKED)
    assert {decode_dna(int(v), 3): int(c) for v, c in zip(kk, cc)} == KAT["kmer3_multiplicity"]["table"]
    assert list(do) == [0, 31]


def test_distribution_weighted_kmer64(oracle):
    """kmergenerator.rs:853-894: 15-mers of the string with a repeated head have weight 2 exactly when they occur twice"""
    s = KAT["seq72_repeat"]
    bases, off = oracle.concat([s.encode()])
    kk, cc, _ = oracle.kmer_distribution(bases, off, A.KMER64BIT, 15, A.FHASH_IDENTITY_RAW)
    assert int(cc.sum()) == len(s) - 14 and set(cc.tolist()) == {1, 2}
    for v, c in zip(kk, cc):
        sub = decode_dna(int(v), 15)
        assert sum(1 for i in range(len(s) - 14) if s[i:i + 15] == sub) == c


def test_nthash_positions_roll_equals_init(oracle):
    """kmo_nthash rolls with the reference's cycle functions (8-bit table); at every position the value must be the init
    value of the k-mer starting there (nthash.rs:333,379), in all three modes, with the strand of :223-227"""
    L = oracle.lib()
    s = KAT["seq80"].encode()
    bases, off = oracle.concat([s])
    buf = np.frombuffer(s, np.uint8)
    for k in (5, 16, 31):
        can, strand = oracle.nthash(bases, off, k, table=A.NTHASH_TABLE_8B)
        fwd, _ = oracle.nthash(bases, off, k, mode=A.NTHASH_FORWARD, table=A.NTHASH_TABLE_8B)
        rc, _ = oracle.nthash(bases, off, k, mode=A.NTHASH_RCOMP, table=A.NTHASH_TABLE_8B)
        for i in range(len(s) - k + 1):
            f, r, st = C.c_uint64(), C.c_uint64(), C.c_uint8()
            h = L.kmo_nthash_canonical_init_8b(buf[i:].ctypes.data, k, C.byref(f), C.byref(r), C.byref(st))
            assert (int(can[i, 0]), int(strand[i]), int(fwd[i, 0]), int(rc[i, 0])) == (h, st.value, f.value, r.value)
            assert f.value == L.kmo_nthash_init_8b(buf[i:].ctypes.data, k) and r.value == L.kmo_nthash_rcomp_init_8b(buf[i:].ctypes.data, k)


def test_nthash_2bit_first_kmer_and_mult(oracle):
    """the 2-bit table: SURVEY 8(a6) derived KAT at position 0, and from_one_hash_val_to_mult_hash (nthash.rs:63-72)"""
    bases, off = oracle.concat([KAT["seq80"].encode()])
    h, strand = oracle.nthash(bases, off, 16, n_hashes=5)
    assert int(h[0, 0]) == 0x684a2ec1114d51c5 and int(strand[0]) == 1
    M = (1 << 64) - 1
    for i in (0, 7, 64):
        for j in range(1, 5):
            t = (int(h[i, 0]) * (j ^ ((16 * 0x90b45d39fb6da1fa) & M))) & M
            assert int(h[i, j]) == t ^ (t >> 27)
    # canonical = min(forward, rcomp); the reverse strand of a sequence gives the same canonical values mirrored
    f, _ = oracle.nthash(bases, off, 16, mode=A.NTHASH_FORWARD)
    r, _ = oracle.nthash(bases, off, 16, mode=A.NTHASH_RCOMP)
    n = 80 - 16 + 1
    assert np.array_equal(h[:n, 0], np.minimum(f[:n, 0], r[:n, 0]))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rcs = "".join(comp[c] for c in reversed(KAT["seq80"]))
    b2, o2 = oracle.concat([rcs.encode()])
    h2, s2 = oracle.nthash(b2, o2, 16)
    assert np.array_equal(h2[:n, 0], h[:n, 0][::-1])
