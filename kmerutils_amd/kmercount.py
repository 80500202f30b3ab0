"""Host-side mirror of the reference's counting interface (src/base/kmercount.rs) on top of libkmu.

  KmerCountT            trait, kmercount.rs:48-59    insert_kmer / get_count / get_nb_distinct / get_nb_unique
  KmerCounter           kmercount.rs:70-123
  KmerCounterPool       kmercount.rs:424-565         one counter per thread -> here one counter per GPU
  count_kmer_threaded_one_to_many   kmercount.rs:881-974
The reference's cuckoo + counting-Bloom pair is approximate and randomised per process; the device table is exact,
i.e. the reference's own contract without its ~3 % false positives (counts saturate at 2^nb_bits - 1).
"""
import numpy as np

from . import _abi as A
from . import lib
from .sketching import _as_arrays, default_context


class KmerCounter:
    """KmerCounter::new(fpr, capacity, nb_bits), kmercount.rs:83-98 (fpr is accepted and ignored: the table is exact)"""

    def __init__(self, kmer_size, capacity, nb_bits=8, fpr=0.03, kmer_type=None, ctx=None):
        self.ctx = ctx or default_context()
        self.kmer_type = kmer_type if kmer_type is not None else A.kmer_type_for_k(kmer_size)
        self.kmer_size = kmer_size
        self._c = lib.Counter(self.ctx, self.kmer_type, kmer_size, nb_bits, capacity)

    def insert_kmer(self, compressed_values):
        """KmerCountT::insert_kmer for one value or an array of `get_compressed_value()` values"""
        v = np.atleast_1d(np.asarray(compressed_values, dtype=np.uint64)).copy()
        self._c.add_kmers(v)

    def insert_reads(self, vseq):
        """every canonical k-mer of every read: kmer.reverse_complement().min(kmer), kmercount.rs:313,938"""
        bases, offsets = _as_arrays(vseq)
        self._c.add_reads(bases, offsets)

    def get_count(self, compressed_values):
        v = np.atleast_1d(np.asarray(compressed_values, dtype=np.uint64)).copy()
        return self._c.query(v)

    def get_nb_distinct(self):
        return self._c.nb_distinct()

    def get_nb_unique(self):
        return self._c.nb_unique()

    def get_above2_count(self, compressed_values):
        """KmerCounter::get_above2_count (kmercount.rs:100-105): the count if the k-mer was seen at least twice, else 0"""
        c = self.get_count(compressed_values)
        return np.where(c >= 2, c, 0).astype(np.uint32)

    def above2_entries(self):
        """(kmer, count) with count >= 2, sorted by k-mer: what dump_kmer_counter writes (kmercount.rs:500-525)"""
        return self._c.dump(2)

    def eliminate_once_kmer(self):
        """kmercount.rs:110-117: forget the k-mers seen once"""
        self._c.eliminate_once()

    def get_count_nb_bits(self):
        return self._c.p.counter_bits

    @property
    def raw(self):
        return self._c


def count_kmer_threaded_one_to_many(seqvec, nb_threads, count_size, kmer_size, capacity=None, ctx=None):
    """kmercount.rs:881-974.  `nb_threads` is accepted for signature compatibility (the GPU needs no key-space
    dispatch inside one device); `count_size` = bits per counter (8 or 16).  Returns a KmerCounter."""
    bases, offsets = _as_arrays(seqvec)
    if capacity is None:
        capacity = max(1024, int(offsets[-1]))
    kc = KmerCounter(kmer_size, capacity, count_size, ctx=ctx)
    kc.insert_reads((bases, offsets))
    return kc


class KmerFilter1:
    """KmerFilter1 (kmercount.rs:985-1089): which 16-mers occur exactly once, and where.  Upstream keeps two cuckoo filters
    (approximate, randomised); here the exact device table answers the same questions.  `insert_reads` stands for the loop
    of filter1_kmer_16b32bit (generate, canonicalise, insert)."""

    def __init__(self, size=16, capacity=1 << 20, ctx=None):
        if size != 16:
            raise ValueError("KmerFilter1 holds Kmer16b32bit")
        self.kmer_size = size
        self.ctx = ctx or default_context()
        self._c = lib.Counter(self.ctx, A.KMER16B32BIT, 16, 8, capacity)

    def insert_kmer16b32bit(self, canonical_values):
        self._c.add_kmers(np.atleast_1d(np.asarray(canonical_values, dtype=np.uint64)).copy())

    def insert_reads(self, vseq):
        bases, offsets = _as_arrays(vseq)
        self._c.add_reads(bases, offsets)

    def get_nb_once(self):
        """once_f.len()"""
        return self._c.nb_unique()

    def once_positions(self, vseq):
        """(kmin, numseq, numkmer) of every occurrence of a once-k-mer in the reads, in file order"""
        bases, offsets = _as_arrays(vseq)
        return self._c.once_positions(bases, offsets)

    def dump_in_file_once_kmer16b32bit(self, fname, seqvec):
        """kmercount.rs:1031-1082; returns the number of k-mers dumped"""
        from . import formats
        k, s, p = self.once_positions(seqvec)
        return formats.dump_once_kmers(fname, np.asarray(k), np.asarray(s), np.asarray(p), 16, 4)


def filter1_kmer_16b32bit(seqvec, ctx=None):
    """kmercount.rs:1093-1123: a KmerFilter1 fed with the canonical 16-mers of every sequence"""
    bases, offsets = _as_arrays(seqvec)
    f = KmerFilter1(16, max(1024, int(offsets[-1])), ctx=ctx)
    f.insert_reads((bases, offsets))
    return f
