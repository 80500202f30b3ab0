"""tests/golden/oracle_signatures.json: the oracle's signatures on the reference's own test strings, frozen.
CPU: the oracle still reproduces them and the reference's statistical assertions hold on the frozen rows.
GPU: the device produces the same rows through the C-ABI."""
import json
import os
import sys

import numpy as np
import pytest

from kmerutils_amd import _abi as A

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "oracle_signatures.json")))
CASES = sorted(FIX["cases"])


def _params(c):
    return A.SketchParams(c["algo"], c["kmer_type"], c["k"], c["m"], c["sig_type"], c["hasher"], c["fhash"], 0, A.MODE_PER_SEQ,
                          A.INPUT_ASCII, A.MEM_HOST, c["flags"])


def _rows(c):
    dt = {A.SIG_U32: np.uint32, A.SIG_F32: np.uint32, A.SIG_U64: np.uint64, A.SIG_F64: np.uint64, A.SIG_U16: np.uint16}[c["sig_type"]]
    return np.array(c["rows"], dtype=dt)


def _bits(rows):
    return rows.view(np.uint32 if rows.dtype.itemsize == 4 else np.uint64) if rows.dtype.kind == "f" else rows


def test_fixture_is_what_the_committed_script_writes():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_oracle_signatures as G
    assert sorted(c[0] for c in G.CASES) == CASES
    for case in G.CASES:
        _, _, _, rows = G.compute(case)
        assert np.array_equal(rows, _rows(FIX["cases"][case[0]])), case[0]


def test_oracle_departures_as_switches_change_no_fixture(monkeypatch):
    """The oracle knowingly departs from the recalled probminhash crate in two measure-zero places (oracle/kmu_oracle.c):
    exact f64 ties go to the smaller key (crate: strict `<`, first in iteration order stays) and the sampler's exp_m1 is a fixed
    Horner form (crate: libm).  Both are switches -- KMO_STRICT_TIES, KMO_LIBM_EXPM1 -- and every ProbMinHash row of the
    committed fixture, plus a batch of ONT-shaped reads at the bench's parameters, is identical under all four combinations
    (call site: seqsketchjaccard.rs:235-240)."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_oracle_signatures as G
    from kmerutils_amd import synth
    from oracle import oracle as O
    bases, off = synth.ont_reads(24, 120_000, 0xC3)
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER64BIT, 31, 200, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    p3 = A.SketchParams(A.ALGO_PROB3, A.KMER32BIT, 8, 200, A.SIG_U32, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    base_rows = None
    for ties in ("0", "1"):
        for libm in ("0", "1"):
            monkeypatch.setenv("KMO_STRICT_TIES", ties)
            monkeypatch.setenv("KMO_LIBM_EXPM1", libm)
            for case in G.CASES:
                if case[2] not in (A.ALGO_PROB3A, A.ALGO_PROB3):
                    continue
                _, _, _, rows = G.compute(case)
                assert np.array_equal(rows, _rows(FIX["cases"][case[0]])), (case[0], ties, libm)
            rows = [O.sketch(bases, off, p), O.sketch(bases, off, p3)]
            if base_rows is None:
                base_rows = rows
            assert all(np.array_equal(a, b) for a, b in zip(rows, base_rows)), (ties, libm)


def test_reference_assertions_hold_on_the_frozen_rows():
    """the thresholds of the reference's tests (seqsketchjaccard.rs:784-785, 850, 902-909, 942-943, 992-1004;
    aautils/setsketchert.rs:1264, 1369) evaluated on the fixture"""
    def jac(rows, i, j):
        return float((rows[i] == rows[j]).mean())
    c = FIX["cases"]
    for name, k in (("pminhasha_kmer_smallb_revcomp", 5), ("pminhasha_k16b32bit", 16), ("pminhash_kmer64bit", 16),
                    ("pminhash_kmer64bit_rand08", 16), ("superminhash_16b32bit_fnv_f64", 16),
                    ("superminhash_16b32bit_nohash_f32", 16)):
        rows = _rows(c[name])
        jac_theo_0 = (40 - k) / (80 - k)
        assert jac(rows, 0, 1) >= 0.75 * jac_theo_0, name
        assert jac(rows, 0, 2) >= 1.0, name          # canonical k-mers: a sequence and its reverse complement
    rows = _rows(c["pminhasha_kmer_smallb_identity"])
    assert jac(rows, 0, 1) >= 0.75 * (40 - 5) / (80 - 5) and jac(rows, 0, 2) <= 0.1
    for name in ("seqaa_probminhash_64bit", "seqaa_probminhash_32bit", "seqaa_superminhash_64bit_f64", "seq_optdensminhash_f64",
                 "seq_revoptdensminhash_f32", "seqaa_optdensminhash_32bit_f64"):   # setsketchert.rs:1124, 1208; aautils :1444
        rows = _rows(c[name])
        assert abs(jac(rows, 0, 1) - 0.5) < 0.1, name


@pytest.fixture(scope="module")
def ctx():
    from kmerutils_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_reproduces_the_frozen_rows(ctx, oracle, name):
    c = FIX["cases"][name]
    bases, off = oracle.concat([s.encode() for s in c["sequences"]])
    got = _bits(np.asarray(ctx.sketch(bases, off, _params(c))))
    assert np.array_equal(got, _rows(c))


def test_frozen_hll_rows_merge_by_maximum():
    """the HLL case holds (seq, its first half, its reverse complement) sketched on canonical k-mers: the reverse complement
    has the same registers, the half has registers no larger"""
    rows = _rows(FIX["cases"]["hyperloglog_u16_default_params"])
    assert np.array_equal(rows[0], rows[2]) and (rows[1] <= rows[0]).all() and rows[0].max() > 0
