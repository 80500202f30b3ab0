"""OptDens / RevOptDens (KMU_ALGO_OPTDENS, KMU_ALGO_REVOPTDENS) through the C-ABI against the oracle: the reference's own
test inputs (src/sketching/setsketchert.rs:1075-1230, src/aautils/setsketchert.rs:1394-1466), then seeded sweeps over
sketch sizes, sequence shapes, input forms and modes.  Bit-exact (float signatures compared as bit patterns)."""
import numpy as np
import pytest

from kmerutils_amd import _abi as A

pytestmark = pytest.mark.gpu

STR1 = b"ATCATGCCCCTTTAGAAAATTTCCGGATCATCGTACGGAGCATGCGTACAACGTCGATGC"
STR2 = b"ATCATGCCCCTTTAGAAAATTTCCGGATCATCATGCCCCTTTAGAAAATTTCCGGATC"
AA1 = b"MTEQIELIKLYSTRILALAAQMPHVGSLDNPDASAMKRSPLCGSKVTVDVIMQNGKITFDGFEVLAPASEYKNRHASILLSLDATAEACASIAAQNSA"
AA2 = b"MTEQIELIKLYSTRILALAAQMPHVGSLDNPDASAMKRSPLCGSKVMTEQIELIKLYSTRILALAAQMPHVGSLDNPDASAMKRSPLCGSKV"


@pytest.fixture(scope="module")
def ctx():
    from kmerutils_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


def bits(x):
    x = np.asarray(x)
    return x.view(np.uint32 if x.dtype.itemsize == 4 else np.uint64)


def params(algo, kmer_type, k, m, sig, fhash=A.FHASH_CANON_INVHASH, hasher=A.HASHER_NOHASH, mode=A.MODE_PER_SEQ, flags=0,
           input_kind=A.INPUT_ASCII):
    return A.SketchParams(algo, kmer_type, k, m, sig, hasher, fhash, 0, mode, input_kind, A.MEM_HOST, flags)


@pytest.mark.parametrize("algo,m", [(A.ALGO_OPTDENS, 800), (A.ALGO_REVOPTDENS, 8000), (A.ALGO_OPTDENS, 8000), (A.ALGO_REVOPTDENS, 800)])
@pytest.mark.parametrize("sig", [A.SIG_F64, A.SIG_F32])
def test_reference_dna_tests(ctx, oracle, algo, m, sig):
    """test_seq_optdensminhash_trait (m = 800) / test_seq_revoptdensminhash_trait (m = 8000): 56 k-mers per sequence, the
    rest of the bins comes from densification; |J - 0.5| < 0.1 and the rows equal the oracle's"""
    bases, off = oracle.concat([STR1, STR2])
    p = params(algo, A.KMER32BIT, 5, m, sig, fhash=A.FHASH_VALUE_MASKED)
    got = np.asarray(ctx.sketch(bases, off, p))
    assert abs(float((got[0] == got[1]).mean()) - 0.5) < 0.1
    assert np.array_equal(bits(got), bits(oracle.sketch(bases, off, p)))
    assert (got < 1.0).all()  # every bin holds some item's r


@pytest.mark.parametrize("sig", [A.SIG_F64, A.SIG_F32])
def test_reference_aa_test(ctx, oracle, sig):
    """test_seqaa_optdensminhash_trait_32bit: KmerAA32bit, k = 5, m = 80"""
    bases, off = oracle.concat([AA1, AA2])
    for algo in (A.ALGO_OPTDENS, A.ALGO_REVOPTDENS):
        p = params(algo, A.KMERAA32BIT, 5, 80, sig, fhash=A.FHASH_VALUE_MASKED)
        got = np.asarray(ctx.sketch(bases, off, p))
        assert abs(float((got[0] == got[1]).mean()) - 0.5) < 0.1
        assert np.array_equal(bits(got), bits(oracle.sketch(bases, off, p)))


def _reads(rng, lens):
    return [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(n)).tobytes() for n in lens]


@pytest.mark.parametrize("algo", [A.ALGO_OPTDENS, A.ALGO_REVOPTDENS])
def test_sweep_sizes_and_shapes(ctx, oracle, algo):
    """reads shorter than k (row stays at the initial value), reads that fill few / most / all bins, several k-mer types,
    both hashers, both index-draw conventions, f32 and f64"""
    rng = np.random.default_rng(7 + algo)
    seqs = _reads(rng, [3, 20, 21, 22, 64, 65, 500, 5000, 40000]) + [b"ACGT" * 300, b"A" * 100]
    bases, off = oracle.concat(seqs)
    for kmer_type, k in ((A.KMER32BIT, 8), (A.KMER16B32BIT, 16), (A.KMER64BIT, 21)):
        for m in (2, 37, 200, 1000, 4096):
            for sig, hasher, flags in ((A.SIG_F64, A.HASHER_NOHASH, 0), (A.SIG_F32, A.HASHER_FNV1A, A.FLAG_RAND08)):
                p = params(algo, kmer_type, k, m, sig, hasher=hasher, flags=flags)
                got, want = bits(ctx.sketch(bases, off, p)), bits(oracle.sketch(bases, off, p))
                assert np.array_equal(got, want), (kmer_type, k, m, sig)
    # packed input (what Sequence::new(raw, 2) holds)
    packed, poff = ctx.pack2b(bases, off)
    p = params(algo, A.KMER64BIT, 21, 300, A.SIG_F64)
    pp = A.SketchParams.from_buffer_copy(p)
    pp.input_kind = A.INPUT_PACKED2
    assert np.array_equal(bits(ctx.sketch(packed, off, pp, packed_offsets=poff)), bits(oracle.sketch(bases, off, p)))


@pytest.mark.parametrize("algo", [A.ALGO_OPTDENS, A.ALGO_REVOPTDENS])
def test_all_seqs_and_hashed_input(ctx, oracle, algo):
    """sketch_compressedkmer_seqs: one set of bins for the whole list; kmu_sketch_hashed: the caller evaluated fhash"""
    import torch
    rng = np.random.default_rng(70 + algo)
    seqs = _reads(rng, [30, 700, 2500, 90, 12000, 5])
    bases, off = oracle.concat(seqs)
    for m in (64, 3000):
        p = params(algo, A.KMER64BIT, 25, m, A.SIG_F64, mode=A.MODE_ALL_SEQS)
        want = bits(oracle.sketch(bases, off, p))
        assert want.shape == (1, m)
        assert np.array_equal(bits(ctx.sketch(bases, off, p)), want)
        db, do = torch.from_numpy(bases.copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
        g = ctx.sketch(db, do, p)
        ctx.synchronize()
        assert np.array_equal(bits(g.cpu().numpy()), want)
    # pre-hashed values (u64 and u32), per sequence and for all
    vals = rng.integers(0, 1 << 62, size=9000, dtype=np.uint64)
    voff = np.array([0, 10, 10, 4000, 9000], np.uint64)
    for mode in (A.MODE_PER_SEQ, A.MODE_ALL_SEQS):
        p = params(algo, A.KMER64BIT, 25, 500, A.SIG_F32, mode=mode)
        assert np.array_equal(bits(ctx.sketch_hashed(vals, voff, p)), bits(oracle.sketch_hashed(vals, voff, p)))
        v32 = (vals & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        p32 = params(algo, A.KMER32BIT, 12, 500, A.SIG_F64, mode=mode)
        assert np.array_equal(bits(ctx.sketch_hashed(v32, voff, p32)), bits(oracle.sketch_hashed(v32, voff, p32)))


def test_genome_sized_sequence_and_errors(ctx, oracle):
    """a sequence beyond 2^20 k-mers is spread over the grid (per sequence and in the all-sequences sketch); an empty
    sequence and a non-ACGT byte are errors as in the oracle; too large a sketch is refused"""
    from kmerutils_amd.lib import KmuError
    rng = np.random.default_rng(99)
    seqs = _reads(rng, [4000, 1_300_000, 250])
    bases, off = oracle.concat(seqs)
    for algo, mode in ((A.ALGO_OPTDENS, A.MODE_PER_SEQ), (A.ALGO_REVOPTDENS, A.MODE_ALL_SEQS), (A.ALGO_REVOPTDENS, A.MODE_PER_SEQ)):
        p = params(algo, A.KMER64BIT, 21, 1024, A.SIG_F64, mode=mode)
        assert np.array_equal(bits(ctx.sketch(bases, off, p)), bits(oracle.sketch(bases, off, p)))
    p = params(A.ALGO_OPTDENS, A.KMER64BIT, 21, 64, A.SIG_F64)
    b2, o2 = oracle.concat([b"ACGTACGTACGTACGTACGTACGTA", b""])
    with pytest.raises(KmuError) as e:
        ctx.sketch(b2, o2, p)
    assert e.value.code == A.E_EMPTY_SEQ
    with pytest.raises(oracle.OracleError):
        oracle.sketch(b2, o2, p)
    b3, o3 = oracle.concat([b"ACGTACGTACGTACGTNCGTACGTACGTACGT"])
    with pytest.raises(KmuError) as e:
        ctx.sketch(b3, o3, p)
    assert e.value.code == A.E_NON_ACGT
    for bad in (params(A.ALGO_OPTDENS, A.KMER64BIT, 21, 64, A.SIG_U64), params(A.ALGO_REVOPTDENS, A.KMER64BIT, 21, 60000, A.SIG_F64)):
        with pytest.raises(KmuError):
            ctx.sketch(bases, off, bad)


# ---- SetSketch registers (KMU_ALGO_HLL, HyperLogLogSketch of src/sketching/setsketchert.rs:640-896) ------------------------
@pytest.mark.parametrize("m,sig", [(256, A.SIG_U16), (4096, A.SIG_U32), (1000, A.SIG_U64)])
def test_hll_registers_match_the_oracle(ctx, oracle, m, sig):
    """per sequence, one for all, caller-hashed values, a genome-sized sequence; default and non-default SetSketchParams"""
    rng = np.random.default_rng(640 + m)
    seqs = _reads(rng, [25, 4000, 60000, 300, 9]) + [b"ACGT" * 500]
    bases, off = oracle.concat(seqs)
    for b, a, q in ((1.001, 20.0, 65534), (1.05, 5.0, 254)):
        if sig == A.SIG_U16 or q < 60000:
            pass
        ctx.set_hll_params(b, a, q)
        oracle.set_hll_params(b, a, q)
        for mode in (A.MODE_PER_SEQ, A.MODE_ALL_SEQS):
            p = params(A.ALGO_HLL, A.KMER64BIT, 21, m, sig, mode=mode)
            want = oracle.sketch(bases, off, p)
            got = np.asarray(ctx.sketch(bases, off, p))
            assert got.dtype == want.dtype and np.array_equal(got, want), (m, sig, b, mode)
            assert want.max() <= q + 1
        vals = rng.integers(0, 1 << 62, size=30000, dtype=np.uint64)
        voff = np.array([0, 50, 50, 30000], np.uint64)
        p = params(A.ALGO_HLL, A.KMER64BIT, 21, m, sig, hasher=A.HASHER_FNV1A, flags=A.FLAG_RAND08)
        assert np.array_equal(np.asarray(ctx.sketch_hashed(vals, voff, p)), oracle.sketch_hashed(vals, voff, p))
    ctx.set_hll_params()
    oracle.set_hll_params()


def test_hll_large_inputs_properties_and_partials(ctx, oracle):
    """a 1.5 Mbase sequence (whole grid), amino acids, the cardinality the registers imply, and mergeability: registers of
    shares merge by maximum to the registers of the whole (kmu_sketch_partial / kmu_sketch_merge_partials)"""
    import math
    rng = np.random.default_rng(641)
    seqs = _reads(rng, [1_500_000, 3000, 200_000])
    bases, off = oracle.concat(seqs)
    m = 4096
    p = params(A.ALGO_HLL, A.KMER64BIT, 25, m, A.SIG_U16, mode=A.MODE_ALL_SEQS)
    want = oracle.sketch(bases, off, p)
    got = np.asarray(ctx.sketch(bases, off, p))
    assert np.array_equal(got, want)
    pp = params(A.ALGO_HLL, A.KMER64BIT, 25, m, A.SIG_U16)
    assert np.array_equal(np.asarray(ctx.sketch(bases, off, pp)), oracle.sketch(bases, off, pp))
    # Ertl's estimator: n = m (1 - 1/b) / (a ln b sum b^-K); distinct canonical 25-mers of random sequences ~ all of them
    K = got[0].astype(np.float64)
    b, a = 1.001, 20.0
    est = m * (1 - 1 / b) / (a * math.log(b) * np.sum(b ** (-K)))
    n_true = np.unique(oracle.kmer_hashes(bases, off, A.KMER64BIT, 25, A.FHASH_CANON_INVHASH)[
        np.concatenate([np.arange(int(off[i]), int(off[i + 1]) - 24) for i in range(3)])]).size
    assert abs(est - n_true) / n_true < 0.08
    parts = np.stack([np.asarray(ctx.sketch_partial(bases, off[a0:b0 + 1].copy(), p)) for a0, b0 in ((0, 1), (1, 3))])
    assert np.array_equal(np.asarray(ctx.sketch_merge_partials(parts, p)), want[0])
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    prot = [rng.choice(aa, size=int(n)).tobytes() for n in (40, 900, 30000)]
    pb, po = oracle.concat(prot)
    pa = params(A.ALGO_HLL, A.KMERAA64BIT, 7, 512, A.SIG_U32, fhash=A.FHASH_VALUE_MASKED)
    assert np.array_equal(np.asarray(ctx.sketch(pb, po, pa)), oracle.sketch(pb, po, pa))
