"""Host-side mirror of the reference's sketching interface, on top of libkmu (the C-ABI).

Names and argument meaning follow the reference so that callers (gsearch-style code, the parity tests) read the same:
  SeqSketcherParams            src/sketcharg.rs:40-78
  SketchAlgo / DataType        src/sketcharg.rs:13-33
  ProbHash3aSketch             src/sketching/setsketchert.rs:85-203      (trait SeqSketcherT :54-80)
  SuperHashSketch              src/sketching/setsketchert.rs:211-336
  SuperHash2Sketch             src/sketching/setsketchert.rs:904-1046
  OptDensHashSketch / RevOptDensHashSketch   src/sketching/setsketchert.rs:343-599
  SeqSketcher                  src/sketching/seqsketchjaccard.rs:117-415 (sketch_probminhash3a :211, _superminhash :328)
  BlockSeqSketcher             src/sketching/seqblocksketch.rs:79-227
A Rust closure `fhash` cannot cross the FFI: pass one of the FHASH_* modes (the closures the reference's own callers
use, include/kmu.h `kmu_fhash`).  Sequences are given as a list of `bytes` (ASCII), or as (bases, offsets) arrays
(numpy on the host, torch tensors on the device).  Errors surface as KmuError where the reference panics.
"""
import numpy as np

from . import _abi as A
from . import lib


class SketchAlgo:
    PROB3A, SUPER, SUPER2, BOTTOMK, PROB3 = A.ALGO_PROB3A, A.ALGO_SUPER, A.ALGO_SUPER2, A.ALGO_BOTTOMK, A.ALGO_PROB3
    OPTDENS, REVOPTDENS, HLL = A.ALGO_OPTDENS, A.ALGO_REVOPTDENS, A.ALGO_HLL


class DataType:
    DNA, AA = 0, 1


class SeqSketcherParams:
    """SeqSketcherParams::new(kmer_size, sketch_size, algo, data_t), src/sketcharg.rs:49-56"""

    def __init__(self, kmer_size, sketch_size, algo=SketchAlgo.PROB3A, data_t=DataType.DNA):
        self.kmer_size, self.sketch_size, self.algo, self.data_t = kmer_size, sketch_size, algo, data_t

    def get_kmer_size(self):
        return self.kmer_size

    def get_sketch_size(self):
        return self.sketch_size

    def get_algo(self):
        return self.algo


def _as_arrays(seqs):
    """list of bytes -> (bases, offsets); arrays pass through"""
    if isinstance(seqs, (tuple, list)) and len(seqs) == 2 and not isinstance(seqs[0], (bytes, bytearray)):
        return seqs[0], seqs[1]
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(bytes(s) for s in seqs), dtype=np.uint8).copy()
    if bases.size == 0:
        bases = np.zeros(16, np.uint8)
    return bases, offsets


_default_ctx = {}


def default_context(device_id=0):
    if device_id not in _default_ctx:
        _default_ctx[device_id] = lib.Context(device_id)
    return _default_ctx[device_id]


class _SketcherBase:
    algo = None

    def __init__(self, params, kmer_type=None, ctx=None):
        self.params = params
        aa = params.data_t == DataType.AA
        self.kmer_type = kmer_type if kmer_type is not None else A.kmer_type_for_k(params.kmer_size, aa)
        self.ctx = ctx or default_context()

    def get_kmer_size(self):
        return self.params.kmer_size

    def get_sketch_size(self):
        return self.params.sketch_size

    def get_algo(self):
        return self.algo

    def _sig_type(self):
        raise NotImplementedError

    def _hasher(self):
        return A.HASHER_NOHASH

    def _params(self, fhash, mode, block_size=0, flags=0):
        return A.SketchParams(self.algo, self.kmer_type, self.params.kmer_size, self.params.sketch_size,
                              self._sig_type(), self._hasher(), fhash, block_size, mode, A.INPUT_ASCII, A.MEM_HOST, flags)

    def sketch_compressedkmer(self, vseq, fhash, flags=0):
        """one signature per sequence, in order (SeqSketcherT::sketch_compressedkmer, setsketchert.rs:66-72)"""
        bases, offsets = _as_arrays(vseq)
        return self.ctx.sketch(bases, offsets, self._params(fhash, A.MODE_PER_SEQ, flags=flags))

    def sketch_compressedkmer_seqs(self, vseq, fhash, flags=0):
        """ONE signature for the whole list (outer length 1), setsketchert.rs:74-79"""
        bases, offsets = _as_arrays(vseq)
        return self.ctx.sketch(bases, offsets, self._params(fhash, A.MODE_ALL_SEQS, flags=flags))


class ProbHash3aSketch(_SketcherBase):
    """type Sig = Kmer::Val (setsketchert.rs:107)"""
    algo = SketchAlgo.PROB3A

    def _sig_type(self):
        return A.SIG_U32 if A.kmer_val_bytes(self.kmer_type) == 4 else A.SIG_U64


class SuperHashSketch(_SketcherBase):
    """type Sig = f32 / f64 (setsketchert.rs:237); NoHashHasher (:267)"""
    algo = SketchAlgo.SUPER

    def __init__(self, params, sig="f64", kmer_type=None, ctx=None):
        super().__init__(params, kmer_type, ctx)
        self.sig = sig

    def _sig_type(self):
        return A.SIG_F32 if self.sig == "f32" else A.SIG_F64


class OptDensHashSketch(SuperHashSketch):
    """OptDensHashSketch<Kmer, S>: one-permutation hashing + optimal densification, type Sig = f32 / f64, NoHashHasher
    (setsketchert.rs:343-463; AA: aautils/setsketchert.rs:482-612)"""
    algo = SketchAlgo.OPTDENS


class RevOptDensHashSketch(SuperHashSketch):
    """RevOptDensHashSketch<Kmer, S>: the same with reverse optimal densification, for sketches larger than the sequences
    (setsketchert.rs:474-599; AA: aautils/setsketchert.rs:616-746)"""
    algo = SketchAlgo.REVOPTDENS


class SetSketchParams:
    """probminhash::setsketcher::SetSketchParams as HyperLogLogSketch::new takes it: b, m, a, q (defaults of the crate)"""

    def __init__(self, b=1.001, m=4096, a=20.0, q=65534):
        self.b, self.m, self.a, self.q = b, m, a, q


class HyperLogLogSketch(_SketcherBase):
    """HyperLogLogSketch<Kmer, S>::new(seq_params, hll_params, hll_threads) (setsketchert.rs:640-896): SetSketch registers,
    type Sig = u16 / u32 / u64.  `hll_threads` (HllSeqsThreading) is accepted for the signature; the device needs no split."""
    algo = SketchAlgo.HLL

    def __init__(self, params, hll_params=None, hll_threads=None, sig="u16", kmer_type=None, ctx=None):
        super().__init__(params, kmer_type, ctx)
        self.hll_params = hll_params or SetSketchParams()
        self.sig = sig

    def _sig_type(self):
        return {"u16": A.SIG_U16, "u32": A.SIG_U32, "u64": A.SIG_U64}[self.sig]

    def _params(self, fhash, mode, block_size=0, flags=0):
        h = self.hll_params
        self.ctx.set_hll_params(h.b, h.a, h.q)
        p = super()._params(fhash, mode, block_size, flags)
        p.sketch_size = h.m  # the sketcher's size is SetSketchParams.m (setsketchert.rs:702-703)
        return p


class SuperHash2Sketch(_SketcherBase):
    """type Sig = u32 / u64, caller-chosen hasher (setsketchert.rs:904-945)"""
    algo = SketchAlgo.SUPER2

    def __init__(self, params, sig="u64", hasher=A.HASHER_NOHASH, kmer_type=None, ctx=None):
        super().__init__(params, kmer_type, ctx)
        self.sig, self.hasher = sig, hasher

    def _sig_type(self):
        return A.SIG_U32 if self.sig == "u32" else A.SIG_U64

    def _hasher(self):
        return self.hasher


class SeqSketcher:
    """SeqSketcher{kmer_size, sketch_size}, src/sketching/seqsketchjaccard.rs:117-140"""

    def __init__(self, kmer_size, sketch_size, ctx=None):
        self.kmer_size, self.sketch_size = kmer_size, sketch_size
        self.ctx = ctx or default_context()

    def get_kmer_size(self):
        return self.kmer_size

    def get_sketch_size(self):
        return self.sketch_size

    def sketch_probminhash3a(self, vseq, fhash, kmer_type=None, flags=0):
        """seqsketchjaccard.rs:211-260: Vec<Vec<Kmer::Val>>, row i <-> sequence i"""
        kt = kmer_type if kmer_type is not None else A.kmer_type_for_k(self.kmer_size)
        sig = A.SIG_U32 if A.kmer_val_bytes(kt) == 4 else A.SIG_U64
        p = A.SketchParams(A.ALGO_PROB3A, kt, self.kmer_size, self.sketch_size, sig, A.HASHER_NOHASH, fhash, 0,
                           A.MODE_PER_SEQ, A.INPUT_ASCII, A.MEM_HOST, flags)
        bases, offsets = _as_arrays(vseq)
        return self.ctx.sketch(bases, offsets, p)

    def sketch_probminhash3(self, vseq, fhash, kmer_type=None, flags=0):
        """seqsketchjaccard.rs:272-319: ProbMinHash3<Kmer::Val, NoHashHasher> over the same k-mer multiset"""
        kt = kmer_type if kmer_type is not None else A.kmer_type_for_k(self.kmer_size)
        sig = A.SIG_U32 if A.kmer_val_bytes(kt) == 4 else A.SIG_U64
        p = A.SketchParams(A.ALGO_PROB3, kt, self.kmer_size, self.sketch_size, sig, A.HASHER_NOHASH, fhash, 0,
                           A.MODE_PER_SEQ, A.INPUT_ASCII, A.MEM_HOST, flags)
        bases, offsets = _as_arrays(vseq)
        return self.ctx.sketch(bases, offsets, p)

    def sketch_superminhash(self, vseq, fhash, sig="f64", kmer_type=None, flags=0):
        """seqsketchjaccard.rs:328-380: SuperMinHash<S, Kmer::Val, fnv::FnvHasher>"""
        kt = kmer_type if kmer_type is not None else A.kmer_type_for_k(self.kmer_size)
        p = A.SketchParams(A.ALGO_SUPER, kt, self.kmer_size, self.sketch_size,
                           A.SIG_F32 if sig == "f32" else A.SIG_F64, A.HASHER_FNV1A, fhash, 0, A.MODE_PER_SEQ,
                           A.INPUT_ASCII, A.MEM_HOST, flags)
        bases, offsets = _as_arrays(vseq)
        return self.ctx.sketch(bases, offsets, p)


class BlockSeqSketcher:
    """BlockSeqSketcher{block_size, kmer_size, sketch_size}, src/sketching/seqblocksketch.rs:79-95 (Kmer32bit only)"""

    def __init__(self, block_size, kmer_size, sketch_size, ctx=None):
        self.block_size, self.kmer_size, self.sketch_size = block_size, kmer_size, sketch_size
        self.ctx = ctx or default_context()

    def blocksketch_sequences(self, vseq, fhash, first_numseq=0):
        """seqblocksketch.rs:152-167.  Returns (rows [n_blocks_total, sketch_size] u32, numseq, numblock): row j is
        BlockSketched{numseq[j], numblock[j], sketch = rows[j]}"""
        bases, offsets = _as_arrays(vseq)
        p = A.SketchParams(A.ALGO_PROB3A, A.KMER32BIT, self.kmer_size, self.sketch_size, A.SIG_U32, A.HASHER_NOHASH,
                           fhash, self.block_size, A.MODE_PER_SEQ, A.INPUT_ASCII, A.MEM_HOST, 0)
        off_h = np.ascontiguousarray(offsets, np.uint64)
        bro = self.ctx.block_layout(off_h, self.block_size)
        rows = self.ctx.sketch(bases, offsets, p, block_row_offsets=bro)
        nb = np.diff(bro.astype(np.int64))
        numseq = np.repeat(np.arange(len(nb), dtype=np.uint32) + first_numseq, nb)
        numblock = np.concatenate([np.arange(k, dtype=np.uint32) for k in nb]) if len(nb) else np.zeros(0, np.uint32)
        return rows, numseq, numblock


# ---- sketches of a range of one sequence (src/sketching/seqminhash.rs) --------------------------------------------------
def _range_kmer_type(kmer_size):
    if kmer_size == 16:
        return A.KMER16B32BIT
    if 9 <= kmer_size <= 15:
        return A.KMER32BIT
    raise ValueError("sketch_seqrange_*: unimplemented kmer_size %d (seqminhash.rs:53, :111)" % kmer_size)


def sketch_seqrange_superminhash(seq, rng, kmer_size, sketch_size, ctx=None):
    """seqminhash.rs:19-62: SuperMinHash<f64, u32, NoHashHasher> over int32_hash(canonical k-mer) of the k-mers inside
    seq[rng[0]:rng[1]] (KmerSeqIterator::set_range).  `seq`: bytes; returns the f64 sketch."""
    ctx = ctx or default_context()
    p = A.SketchParams(A.ALGO_SUPER, _range_kmer_type(kmer_size), kmer_size, sketch_size, A.SIG_F64, A.HASHER_NOHASH,
                       A.FHASH_CANON_INVHASH, 0, A.MODE_PER_SEQ, A.INPUT_ASCII, A.MEM_HOST, 0)
    bases, offsets = _as_arrays([bytes(seq)[rng[0]:rng[1]]])
    return np.asarray(ctx.sketch(bases, offsets, p))[0]


def sketch_seqrange_minhash(seq, rng, kmer_size, sketch_size, ctx=None):
    """seqminhash.rs:65-119: MinHashCount<u32, NoHashHasher> (bottom-k with u16 multiplicities) over the same values.
    Returns (hashes ascending, counts): the HashCount list of get_sketchcount, sorted."""
    ctx = ctx or default_context()
    p = A.SketchParams(A.ALGO_BOTTOMK, _range_kmer_type(kmer_size), kmer_size, sketch_size, A.SIG_U64, A.HASHER_NOHASH,
                       A.FHASH_CANON_INVHASH, 0, A.MODE_PER_SEQ, A.INPUT_ASCII, A.MEM_HOST, 0)
    bases, offsets = _as_arrays([bytes(seq)[rng[0]:rng[1]]])
    h, c = ctx.sketch(bases, offsets, p, want_counts=True)
    h, c = np.asarray(h)[0], np.asarray(c)[0]
    n = int((h != np.uint64(0xFFFFFFFFFFFFFFFF)).sum())
    return h[:n], c[:n]


# ---- what the callers do with the signatures next: all on the device through libkmu -----------
def probminhash_get_jaccard_objects(siga, sigb, ctx=None):
    """probminhash_get_jaccard_objects(siga, sigb) -> (jp, Some(common objects) | None), src/sketching/
    seqsketchjaccard.rs:86-108.  The count of equal slots comes from kmu_sig_equal_pairs."""
    ctx = ctx or default_context()
    a = np.ascontiguousarray(siga).reshape(1, -1)
    b = np.ascontiguousarray(sigb).reshape(1, -1)
    assert a.shape == b.shape
    z = np.zeros(1, np.uint32)
    inter = int(ctx.sig_equal_pairs(a, b, z, z)[0])
    jp = inter / a.shape[1]
    if jp > 0.0:
        return jp, a[0][a[0] == b[0]].tolist()
    return 0.0, None


def jaccard_matrix(sig_a, sig_b=None, ctx=None):
    """all-pairs slot-equality Jaccard of signature rows (numpy or device tensors): kmu_sig_equal_matrix / m"""
    ctx = ctx or default_context()
    sig_b = sig_a if sig_b is None else sig_b
    eq = ctx.sig_equal_matrix(sig_a, sig_b)
    m = sig_a.shape[1]
    return (eq.astype(np.float64) if isinstance(eq, np.ndarray) else eq.double()) / m


class DistBlockSketched:
    """Distance<BlockSketched>, src/sketching/seqblocksketch.rs:419-431: 1.0 inside one sequence, else the fraction of
    differing slots.  `eval_pairs` takes block rows + their numseq (BlockSeqSketcher.blocksketch_sequences) and index
    pairs."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()

    def eval_pairs(self, rows, numseq, ia, ib):
        ia = np.ascontiguousarray(ia, np.uint32)
        ib = np.ascontiguousarray(ib, np.uint32)
        rows = np.ascontiguousarray(rows)
        eq = self.ctx.sig_equal_pairs(rows, rows, ia, ib).astype(np.float32)
        m = rows.shape[1]
        d = (m - eq) / np.float32(m)
        numseq = np.asarray(numseq)
        d[numseq[ia] == numseq[ib]] = 1.0
        return d


def minhash_distance(hashes_a, hashes_b, ia, ib, ctx=None):
    """minhash_distance / mininvhash_distance (src/sketching/minhash.rs:134-190, :295-340) for pairs of bottom-k rows:
    returns (containment, jaccard, common, total) arrays -- the fields of MinHashDist."""
    ctx = ctx or default_context()
    ia = np.ascontiguousarray(ia, np.uint32)
    ib = np.ascontiguousarray(ib, np.uint32)
    r = np.asarray(ctx.minhash_distance_pairs(np.ascontiguousarray(hashes_a, np.uint64),
                                              np.ascontiguousarray(hashes_b, np.uint64), ia, ib)).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return r[:, 0] / r[:, 2], r[:, 0] / r[:, 1], r[:, 0].astype(np.uint64), r[:, 1].astype(np.uint64)
