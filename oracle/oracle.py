"""ctypes loader for the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  It is the checker /
the reported CPU baseline, never the thing shipped: kmerutils_amd never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from kmerutils_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
# KMU_ORACLE_SO: another build of the same source (`make sanitize`: AddressSanitizer + UBSan under the CPU test suite)
_SO = os.environ.get("KMU_ORACLE_SO") or os.path.join(_HERE, "_build", "libkmu_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "kmu_oracle.c")
        if not os.environ.get("KMU_ORACLE_SO") and (not os.path.exists(_SO) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO))):
            build()
        _lib = C.CDLL(_SO)
        L = _lib
        u8p, u64p, u32p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p
        L.kmo_encode2b.argtypes = [C.c_uint8]; L.kmo_encode2b.restype = C.c_int
        L.kmo_encode_aa.argtypes = [C.c_uint8]; L.kmo_encode_aa.restype = C.c_int
        L.kmo_count_non_acgt.argtypes = [vp, C.c_uint64]; L.kmo_count_non_acgt.restype = C.c_uint64
        L.kmo_pack2b.argtypes = [vp, C.c_uint64, vp]; L.kmo_pack2b.restype = C.c_int64
        L.kmo_pack2b_filtered.argtypes = [vp, C.c_uint64, vp]; L.kmo_pack2b_filtered.restype = C.c_int64
        L.kmo_get_base.argtypes = [vp, C.c_uint64]; L.kmo_get_base.restype = C.c_uint8
        for f in ("kmo_kmer_build",):
            getattr(L, f).argtypes = [C.c_int, C.c_uint64, C.c_int]; getattr(L, f).restype = C.c_uint64
        L.kmo_kmer_push.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_uint8]; L.kmo_kmer_push.restype = C.c_uint64
        L.kmo_kmer_revcomp.argtypes = [C.c_int, C.c_uint64, C.c_int]; L.kmo_kmer_revcomp.restype = C.c_uint64
        L.kmo_kmer_value.argtypes = [C.c_int, C.c_uint64]; L.kmo_kmer_value.restype = C.c_uint64
        L.kmo_kmer_less.argtypes = [C.c_int, C.c_uint64, C.c_uint64]; L.kmo_kmer_less.restype = C.c_int
        L.kmo_int32_hash.argtypes = [C.c_uint32]; L.kmo_int32_hash.restype = C.c_uint32
        L.kmo_int64_hash.argtypes = [C.c_uint64]; L.kmo_int64_hash.restype = C.c_uint64
        L.kmo_minimizer_hash.argtypes = [C.c_uint64, C.c_int]; L.kmo_minimizer_hash.restype = C.c_uint32
        L.kmo_minimizer_owner.argtypes = [C.c_uint64, C.c_int, C.c_uint32]; L.kmo_minimizer_owner.restype = C.c_uint32
        L.kmo_minimizer_owners.argtypes = [vp, C.c_uint64, C.c_int, C.c_uint32, vp]; L.kmo_minimizer_owners.restype = None
        L.kmo_superkmer_expand.argtypes = [vp, C.c_uint64, C.c_int, vp, C.POINTER(C.c_int)]; L.kmo_superkmer_expand.restype = C.c_uint64
        L.kmo_unif01_f64.argtypes = [vp]; L.kmo_unif01_f64.restype = C.c_double
        L.kmo_nohash_finish.argtypes = [C.c_uint64, C.c_int]; L.kmo_nohash_finish.restype = C.c_uint64
        L.kmo_fnv1a.argtypes = [C.c_uint64, C.c_int]; L.kmo_fnv1a.restype = C.c_uint64
        L.kmo_nthash_init_8b.argtypes = [vp, C.c_int]; L.kmo_nthash_init_8b.restype = C.c_uint64
        L.kmo_nthash_cycle_8b.argtypes = [C.c_uint64, C.c_int, C.c_uint8, C.c_uint8]
        L.kmo_nthash_cycle_8b.restype = C.c_uint64
        L.kmo_nthash_canonical_init_8b.argtypes = [vp, C.c_int, u64p, u64p, u8p]
        L.kmo_nthash_canonical_init_8b.restype = C.c_uint64
        L.kmo_nthash_canonical_cycle_8b.argtypes = [C.c_int, C.c_uint8, C.c_uint8, u64p, u64p, u8p]
        L.kmo_nthash_canonical_cycle_8b.restype = C.c_uint64
        L.kmo_nthash_canonical_2b.argtypes = [C.c_uint64, C.c_int, u64p, u64p, u8p]
        L.kmo_nthash_canonical_2b.restype = C.c_uint64
        L.kmo_nthash_mult.argtypes = [C.c_uint64, vp, C.c_int]; L.kmo_nthash_mult.restype = None
        L.kmo_xoshiro_seed.argtypes = [C.c_uint64, vp]; L.kmo_xoshiro_seed.restype = None
        L.kmo_xoshiro_next.argtypes = [vp]; L.kmo_xoshiro_next.restype = C.c_uint64
        L.kmo_kmer_hashes.argtypes = [C.POINTER(A.HashParams), vp, vp, vp, C.c_uint32, vp]
        L.kmo_kmer_hashes.restype = C.c_int
        L.kmo_kmer_hashes_range.argtypes = [C.POINTER(A.HashParams), vp, vp, vp, C.c_uint32, vp, vp, vp]
        L.kmo_kmer_hashes_range.restype = C.c_int
        L.kmo_kmer_distribution.argtypes = [C.POINTER(A.HashParams), vp, vp, vp, C.c_uint32, vp, vp, C.c_uint64, vp, u64p]
        L.kmo_kmer_distribution.restype = C.c_int
        L.kmo_nthash.argtypes = [C.POINTER(A.NthashParams), vp, vp, vp, C.c_uint32, vp, vp]
        L.kmo_nthash.restype = C.c_int
        L.kmo_nthash_rcomp_init_8b.argtypes = [vp, C.c_int]; L.kmo_nthash_rcomp_init_8b.restype = C.c_uint64
        L.kmo_sketch.argtypes = [C.POINTER(A.SketchParams), vp, vp, vp, C.c_uint32, vp, vp, vp]
        L.kmo_sketch.restype = C.c_int
        L.kmo_sketch_hashed.argtypes = [C.POINTER(A.SketchParams), vp, vp, C.c_uint32, vp, vp]
        L.kmo_sketch_hashed.restype = C.c_int
        L.kmo_probminhash3a.argtypes = [vp, vp, C.c_uint64, C.c_int, C.c_int, C.c_uint32, vp, vp]
        L.kmo_probminhash3a.restype = C.c_int
        L.kmo_count_create.argtypes = [C.POINTER(A.CountParams)]; L.kmo_count_create.restype = vp
        L.kmo_count_destroy.argtypes = [vp]; L.kmo_count_destroy.restype = None
        L.kmo_count_add_reads.argtypes = [vp, vp, vp, C.c_uint32]; L.kmo_count_add_reads.restype = C.c_int
        L.kmo_count_add_kmers.argtypes = [vp, vp, C.c_uint64]; L.kmo_count_add_kmers.restype = C.c_int
        L.kmo_count_query.argtypes = [vp, vp, C.c_uint64, vp]; L.kmo_count_query.restype = C.c_int
        L.kmo_count_nb_distinct.argtypes = [vp]; L.kmo_count_nb_distinct.restype = C.c_uint64
        L.kmo_count_nb_unique.argtypes = [vp]; L.kmo_count_nb_unique.restype = C.c_uint64
        L.kmo_count_dump.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint64, u64p]; L.kmo_count_dump.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(A.STATUS_NAMES.get(code, str(code)))
        self.code = code


def concat(seqs):
    """list of bytes -> (bases uint8[total], offsets uint64[n+1])"""
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if len(seqs) else np.zeros(0, np.uint8)
    if bases.size == 0:
        bases = np.zeros(1, np.uint8)
    return bases, offsets


def pack2b(raw):
    raw = np.frombuffer(bytes(raw), dtype=np.uint8)
    out = np.zeros((raw.size + 3) // 4 + 1, np.uint8)
    n = lib().kmo_pack2b(_p(raw) if raw.size else None, raw.size, _p(out))
    if n < 0:
        raise OracleError(A.E_NON_ACGT)
    return out[:n].copy()


def kmer_hashes(bases, offsets, kmer_type, k, fhash, input_kind=A.INPUT_ASCII, packed_offsets=None):
    hp = A.HashParams(kmer_type, k, fhash, input_kind, A.MEM_HOST, 0)
    out = np.zeros(max(int(offsets[-1]), 1), dtype=np.uint64)
    rc = lib().kmo_kmer_hashes(C.byref(hp), _p(bases), _p(offsets), _p(packed_offsets), len(offsets) - 1, _p(out))
    if rc:
        raise OracleError(rc)
    return out


def kmer_hashes_range(bases, offsets, kmer_type, k, fhash, range_begin, range_end, input_kind=A.INPUT_ASCII,
                      packed_offsets=None):
    hp = A.HashParams(kmer_type, k, fhash, input_kind, A.MEM_HOST, 0)
    out = np.zeros(max(int(offsets[-1]), 1), dtype=np.uint64)
    rb, re = np.ascontiguousarray(range_begin, np.uint64), np.ascontiguousarray(range_end, np.uint64)
    rc = lib().kmo_kmer_hashes_range(C.byref(hp), _p(bases), _p(offsets), _p(packed_offsets), len(offsets) - 1, _p(rb), _p(re),
                                     _p(out))
    if rc:
        raise OracleError(rc)
    return out


def kmer_distribution(bases, offsets, kmer_type, k, fhash=A.FHASH_IDENTITY_RAW, input_kind=A.INPUT_ASCII,
                      packed_offsets=None):
    """(values ascending per sequence, multiplicities, dist_offsets)"""
    hp = A.HashParams(kmer_type, k, fhash, input_kind, A.MEM_HOST, 0)
    n = len(offsets) - 1
    cnt = C.c_uint64(0)
    rc = lib().kmo_kmer_distribution(C.byref(hp), _p(bases), _p(offsets), _p(packed_offsets), n, None, None, 0, None, C.byref(cnt))
    if rc:
        raise OracleError(rc)
    kk, cc, do = np.zeros(max(cnt.value, 1), np.uint64), np.zeros(max(cnt.value, 1), np.uint32), np.zeros(n + 1, np.uint64)
    rc = lib().kmo_kmer_distribution(C.byref(hp), _p(bases), _p(offsets), _p(packed_offsets), n, _p(kk), _p(cc), cnt.value, _p(do),
                                     C.byref(cnt))
    if rc:
        raise OracleError(rc)
    return kk[:cnt.value], cc[:cnt.value], do


def nthash(bases, offsets, k, n_hashes=1, mode=A.NTHASH_CANONICAL, table=A.NTHASH_TABLE_2B, input_kind=A.INPUT_ASCII,
           packed_offsets=None):
    np_ = A.NthashParams(k, table, mode, n_hashes, input_kind, A.MEM_HOST)
    nb = max(int(offsets[-1]), 1)
    out = np.zeros((nb, n_hashes), np.uint64)
    strand = np.zeros(nb, np.uint8)
    rc = lib().kmo_nthash(C.byref(np_), _p(bases), _p(offsets), _p(packed_offsets), len(offsets) - 1, _p(out), _p(strand))
    if rc:
        raise OracleError(rc)
    return out, strand


def block_layout(offsets, block_size):
    L = np.diff(offsets.astype(np.int64))
    nb = (L + block_size - 1) // block_size
    out = np.zeros(len(offsets), np.uint64)
    out[1:] = np.cumsum(nb)
    return out


def sketch(bases, offsets, params, packed_offsets=None, want_counts=False):
    n = len(offsets) - 1
    m = params.sketch_size
    bro = None
    if params.mode == A.MODE_ALL_SEQS:
        rows = 1
    elif params.block_size > 0:
        bro = block_layout(offsets, params.block_size)
        rows = int(bro[-1])
    else:
        rows = n
    sig = np.zeros((rows, m), dtype=A.SIG_NP[params.sig_type])
    counts = np.zeros((rows, m), np.uint32) if want_counts else None
    p = A.SketchParams.from_buffer_copy(params)
    p.mem = A.MEM_HOST
    rc = lib().kmo_sketch(C.byref(p), _p(bases), _p(offsets), _p(packed_offsets), n, _p(bro), _p(sig), _p(counts))
    if rc:
        raise OracleError(rc)
    return (sig, counts) if want_counts else sig


def sketch_hashed(hashed, offsets, params, want_counts=False):
    n = len(offsets) - 1
    rows = 1 if params.mode == A.MODE_ALL_SEQS else n
    sig = np.zeros((rows, params.sketch_size), dtype=A.SIG_NP[params.sig_type])
    counts = np.zeros((rows, params.sketch_size), np.uint32) if want_counts else None
    rc = lib().kmo_sketch_hashed(C.byref(params), _p(hashed), _p(offsets), n, _p(sig), _p(counts))
    if rc:
        raise OracleError(rc)
    return (sig, counts) if want_counts else sig


def probminhash3a(keys, weights, key_bytes, m, flags=0):
    keys = np.ascontiguousarray(keys, np.uint64)
    weights = np.ascontiguousarray(weights, np.float64)
    sig = np.zeros(m, np.uint64)
    h = np.zeros(m, np.float64)
    rc = lib().kmo_probminhash3a(_p(keys), _p(weights), keys.size, key_bytes, m, flags, _p(sig), _p(h))
    if rc:
        raise OracleError(rc)
    return sig, h


class Counter:
    def __init__(self, kmer_type, k, counter_bits=8, capacity_hint=1024):
        self.p = A.CountParams(kmer_type, k, counter_bits, 0, capacity_hint)
        self.h = lib().kmo_count_create(C.byref(self.p))
        if not self.h:
            raise OracleError(A.E_BAD_ARG)

    def __del__(self):
        if getattr(self, "h", None):
            lib().kmo_count_destroy(self.h)
            self.h = None

    def add_reads(self, bases, offsets):
        rc = lib().kmo_count_add_reads(self.h, _p(bases), _p(offsets), len(offsets) - 1)
        if rc:
            raise OracleError(rc)

    def add_kmers(self, canon):
        canon = np.ascontiguousarray(canon, np.uint64)
        lib().kmo_count_add_kmers(self.h, _p(canon), canon.size)

    def query(self, canon):
        canon = np.ascontiguousarray(canon, np.uint64)
        out = np.zeros(canon.size, np.uint32)
        lib().kmo_count_query(self.h, _p(canon), canon.size, _p(out))
        return out

    def nb_distinct(self):
        return int(lib().kmo_count_nb_distinct(self.h))

    def nb_unique(self):
        return int(lib().kmo_count_nb_unique(self.h))

    def once_positions(self, bases, offsets):
        """(kmin, numseq, numkmer) of the k-mer occurrences whose canonical k-mer was seen exactly once, in file order"""
        f = lib().kmo_count_once_positions
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        n = C.c_uint64(0)
        rc = f(self.h, _p(bases), _p(offsets), len(offsets) - 1, None, None, None, C.byref(n))
        if rc:
            raise OracleError(rc)
        k, s, p = np.zeros(max(n.value, 1), np.uint64), np.zeros(max(n.value, 1), np.uint32), np.zeros(max(n.value, 1), np.uint32)
        f(self.h, _p(bases), _p(offsets), len(offsets) - 1, _p(k), _p(s), _p(p), C.byref(n))
        return k[:n.value], s[:n.value], p[:n.value]

    def dump(self, min_count=2):
        n = C.c_uint64(0)
        lib().kmo_count_dump(self.h, min_count, None, None, 0, C.byref(n))
        k = np.zeros(max(n.value, 1), np.uint64)
        c = np.zeros(max(n.value, 1), np.uint32)
        lib().kmo_count_dump(self.h, min_count, _p(k), _p(c), n.value, C.byref(n))
        return k[:n.value], c[:n.value]


# ---- signature comparison ----
def sig_equal_count(a, b):
    """number of equal slots of two signature rows (raw 4- or 8-byte words)"""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.dtype.itemsize == b.dtype.itemsize and a.size == b.size
    f = lib().kmo_sig_equal_count
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    f.restype = C.c_uint32
    return int(f(_p(a), _p(b), a.size, a.dtype.itemsize))


def minhash_distance(s1, s2):
    """(common, total, i) of two ascending hash lists (valid entries only)"""
    s1 = np.ascontiguousarray(s1, np.uint64)
    s2 = np.ascontiguousarray(s2, np.uint64)
    out = np.zeros(3, np.uint32)
    f = lib().kmo_minhash_distance
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    f.restype = None
    f(_p(s1), s1.size, _p(s2), s2.size, _p(out))
    return tuple(int(x) for x in out)


# ---- ingest ----
def ingest_fastq(text, fmt="fastq"):
    """(bases, offsets, info dict, record_index) of the ACGT-only reads of a 4-line FASTQ text (fmt "fasta": a FASTA text,
    "fastx": by the first byte), in file order"""
    text = np.frombuffer(bytes(text), np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
    f = getattr(lib(), "kmo_ingest_" + fmt)
    f.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f.restype = C.c_int
    info = np.zeros(6, np.uint64)
    rc = f(_p(text) if text.size else None, text.size, None, None, None, _p(info))
    if rc:
        raise OracleError(rc)
    bases = np.zeros(max(int(info[2]), 1), np.uint8)
    offs = np.zeros(int(info[1]) + 1, np.uint64)
    idx = np.zeros(max(int(info[1]), 1), np.uint32)
    rc = f(_p(text) if text.size else None, text.size, _p(bases), _p(offs), _p(idx), _p(info))
    if rc:
        raise OracleError(rc)
    names = ("n_records", "n_kept", "kept_bases", "n_bases", "nb_bad_bases", "nb_bad_reads")
    return bases[:int(info[2])], offs, dict(zip(names, (int(x) for x in info))), idx[:int(info[1])]


def ingest_fasta(text):
    return ingest_fastq(text, "fasta")


def ingest_fastx(text):
    return ingest_fastq(text, "fastx")


def set_hll_params(b=1.001, a=20.0, q=65534):
    """SetSketchParams of the following KMU_ALGO_HLL sketches"""
    f = lib().kmo_set_hll_params
    f.argtypes = [C.c_double, C.c_double, C.c_uint32]
    f.restype = None
    f(b, a, q)


def minimizer_owners(canon_kmers, k, n_parts):
    """the minimizer owner of Kmer64bit values of k bases (the product's owner function between GPUs, restated base by base)"""
    v = np.ascontiguousarray(canon_kmers, np.uint64)
    out = np.zeros(max(v.size, 1), np.uint32)
    lib().kmo_minimizer_owners(_p(v), v.size, k, n_parts, _p(out))
    return out[:v.size]


def superkmer_expand(records, k):
    """canonical k-mers held by [n, 3] uint32 super-k-mer records, in record order; also whether every record is zero-padded"""
    r = np.ascontiguousarray(records, np.uint32).reshape(-1, 3)
    out = np.zeros(max(16 * r.shape[0], 1), np.uint64)
    clean = C.c_int(1)
    n = lib().kmo_superkmer_expand(_p(r), r.shape[0], k, _p(out), C.byref(clean))
    return out[:n], bool(clean.value)
