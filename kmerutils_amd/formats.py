"""The reference's on-disk dumps, written and read on the host like upstream: signatures,
block signatures, and the (k-mer, count) dump of the counter.  Byte layouts follow the reference's WRITERS (they are
what the Julia companion reads); where upstream's own reader disagrees with its writer this is noted.

  signatures        create_signature_dump / dump_signatures_block_u32   src/sketching/seqsketchjaccard.rs:385-414, 572-583
                    SigSketchFileReader                                   src/sketching/seqsketchjaccard.rs:586-712
  block signatures  create_signature_dump / dump_blocks / BlockSketched::dump
                                                                          src/sketching/seqblocksketch.rs:33, 59-65, 172-226
  k-mer counts      KmerCounterPool::dump_kmer_counter (COUNTER_MULTIPLE) src/base/kmercount.rs:35-41, 467-531
                    Kmer*::dump                                           src/base/kmer64bit.rs:98-104, kmer32bit.rs:141-144

All integers little-endian (`to_le_bytes`, or `transmute` on the little-endian hosts the reference runs on).
"""
import json
import os
import struct

import numpy as np

MAGIC_SIG_DUMP = 0xCEABEADD       # seqsketchjaccard.rs:570
MAGIC_BLOCKSIG_DUMP = 0xCEABBADD  # seqblocksketch.rs:33
COUNTER_UNIQUE = 0xCEA2BBDD       # kmercount.rs:35
COUNTER_MULTIPLE = 0xCEA2BBFF     # kmercount.rs:41


# ---- signatures of whole sequences -----------------------------------------------------------------------------------
def create_signature_dump(fname, sketch_size, kmer_size):
    """SeqSketcher::create_signature_dump: magic, sig_size = 4, sketch_size, kmer_size -- four u32.  Returns the open file."""
    f = open(fname, "wb")
    f.write(struct.pack("<IIII", MAGIC_SIG_DUMP, 4, sketch_size, kmer_size))
    return f


def dump_signatures_block_u32(signatures, out):
    """dump_signatures_block_u32: the rows, one u32 after the other"""
    a = np.ascontiguousarray(signatures)
    if a.dtype.itemsize != 4:
        raise ValueError("the reference dumps Vec<u32> signatures only (sig_size = 4)")
    out.write(a.astype("<u4", copy=False).tobytes())


class SigSketchFileReader:
    """SigSketchFileReader::new / next.  (Upstream's `next` reads the bytes of a row but returns an empty Vec,
    seqsketchjaccard.rs:690-708; this reader returns the row.)"""

    def __init__(self, fname):
        self.f = open(fname, "rb")
        head = self.f.read(16)
        if len(head) < 4:
            raise IOError("SigSketchFileReader could no read magic")
        magic = struct.unpack("<I", head[:4])[0]
        if magic != MAGIC_SIG_DUMP:
            raise IOError("file is not a dump of signature")
        if len(head) < 16:
            raise IOError("SigSketchFileReader could no read sketch_size")
        self.sig_size, self.sketch_size, self.kmer_size = struct.unpack("<III", head[4:])
        if self.sig_size != 4:
            raise IOError("SigSketchFileReader , sig_size != 4 not yet implemented")

    def get_kmer_size(self):
        return self.kmer_size

    def get_signature_length(self):
        return self.sketch_size

    def get_signature_size(self):
        return self.sig_size

    def next(self):
        buf = self.f.read(4 * self.sketch_size)
        if len(buf) < 4 * self.sketch_size:
            return None
        return np.frombuffer(buf, "<u4").copy()

    def read_all(self):
        rows = np.frombuffer(self.f.read(), "<u4")
        return rows[:rows.size - rows.size % self.sketch_size].reshape(-1, self.sketch_size).copy()


# ---- block signatures ------------------------------------------------------------------------------------------------
def create_block_signature_dump(fname, sketch_size, kmer_size, block_size):
    """BlockSeqSketcher::create_signature_dump: magic u32, sig_size as ONE byte (the field is a u8, seqblocksketch.rs:80,
    220: "dump 17 bytes"), sketch_size, kmer_size, block_size u32."""
    f = open(fname, "wb")
    f.write(struct.pack("<IBIII", MAGIC_BLOCKSIG_DUMP, 4, sketch_size, kmer_size, block_size))
    return f


def dump_blocks(out, rows, numseq, numblock):
    """BlockSeqSketcher::dump_blocks: per sequence `numseq` u32, `nbblock` u32, then for every block
    BlockSketched::dump = numseq u32, numblock u32, sketch[u32; m].  `rows`, `numseq`, `numblock` as returned by
    BlockSeqSketcher.blocksketch_sequences (blocks of one sequence are consecutive)."""
    rows = np.ascontiguousarray(rows).astype("<u4", copy=False)
    numseq = np.asarray(numseq, np.uint32)
    numblock = np.asarray(numblock, np.uint32)
    i, n = 0, len(numseq)
    while i < n:
        j = i
        while j < n and numseq[j] == numseq[i]:
            j += 1
        out.write(struct.pack("<II", int(numseq[i]), j - i))
        for b in range(i, j):
            out.write(struct.pack("<II", int(numseq[b]), int(numblock[b])))
            out.write(rows[b].tobytes())
        i = j


class SigBlockSketchFileReader:
    """Reader of what dump_blocks writes.  (Upstream's SigBlockSketchFileReader expects a 4-byte sig_size and rows
    without the per-block (numseq, numblock) words, seqblocksketch.rs:273-288, 357-402: it cannot read upstream's own
    dumps; this reader follows the writer.)"""

    def __init__(self, fname):
        self.f = open(fname, "rb")
        head = self.f.read(17)
        if len(head) < 17 or struct.unpack("<I", head[:4])[0] != MAGIC_BLOCKSIG_DUMP:
            raise IOError("file is not a dump of signature")
        self.sig_size, self.sketch_size, self.kmer_size, self.block_size = struct.unpack("<BIII", head[4:])
        if self.sig_size != 4:
            raise IOError("SigBlockSketchFileReader , sig_size != 4 not yet implemented")

    def next(self):
        """(numseq, [(numblock, row), ...]) of the next sequence, or None"""
        head = self.f.read(8)
        if len(head) < 8:
            return None
        numseq, nbblock = struct.unpack("<II", head)
        blocks = []
        for _ in range(nbblock):
            ns, nb = struct.unpack("<II", self.f.read(8))
            assert ns == numseq
            blocks.append((nb, np.frombuffer(self.f.read(4 * self.sketch_size), "<u4").copy()))
        return numseq, blocks


# ---- k-mer counts ------------------------------------------------------------------------------------------------------
def dump_kmer_counter(fname, kmers, counts, kmer_size, val_bytes):
    """KmerCounterPool::dump_kmer_counter: COUNTER_MULTIPLE u32, kmer_size u8, nb_bytes_by_count u8 (= 1), number of
    k-mers u64, then per k-mer `Kmer::dump` + count u8.  `Kmer::dump`: Kmer64bit = size byte + u64 value
    (kmer64bit.rs:98-104); Kmer32bit / Kmer16b32bit = the u32 word `.0` (for Kmer32bit that word carries k in its top
    nibble, kmer32bit.rs:141-144).  `kmers` are canonical values, `counts` >= 2 (kmu_count_dump(min_count = 2)).
    Upstream writes records in order of first occurrence; the order carries no meaning for a reader."""
    kmers = np.asarray(kmers, np.uint64)
    counts = np.minimum(np.asarray(counts, np.uint64), 255).astype(np.uint8)
    with open(fname, "wb") as f:
        f.write(struct.pack("<IBBQ", COUNTER_MULTIPLE, kmer_size, 1, len(kmers)))
        if val_bytes == 8:
            rec = np.zeros(len(kmers), dtype=[("k", "u1"), ("v", "<u8"), ("c", "u1")])
            rec["k"], rec["v"], rec["c"] = kmer_size, kmers, counts
        else:
            words = kmers.astype(np.uint32)
            if kmer_size <= 14:
                words = words | np.uint32(kmer_size << 28)
            rec = np.zeros(len(kmers), dtype=[("v", "<u4"), ("c", "u1")])
            rec["v"], rec["c"] = words, counts
        f.write(rec.tobytes())
    return len(kmers)


def dump_once_kmers(fname, kmers, numseq, numkmer, kmer_size, val_bytes=4):
    """KmerFilter1::dump_in_file_once_kmer16b32bit (kmercount.rs:1031-1082): COUNTER_UNIQUE u32, kmer_size u8, number of
    k-mers u64, then per record `Kmer::dump` (the u32 word for Kmer16b32bit / Kmer32bit, size byte + u64 for Kmer64bit),
    numseq u32, numkmer u32.  Records as returned by kmu_count_once_positions (file order)."""
    kmers = np.asarray(kmers, np.uint64)
    with open(fname, "wb") as f:
        f.write(struct.pack("<IBQ", COUNTER_UNIQUE, kmer_size, len(kmers)))
        if val_bytes == 8:
            rec = np.zeros(len(kmers), dtype=[("k", "u1"), ("v", "<u8"), ("s", "<u4"), ("p", "<u4")])
            rec["k"] = kmer_size
        else:
            rec = np.zeros(len(kmers), dtype=[("v", "<u4"), ("s", "<u4"), ("p", "<u4")])
        words = kmers if val_bytes == 8 else kmers.astype(np.uint32)
        if val_bytes == 4 and kmer_size <= 14:
            words = words | np.uint32(kmer_size << 28)
        rec["v"], rec["s"], rec["p"] = words, np.asarray(numseq, np.uint32), np.asarray(numkmer, np.uint32)
        f.write(rec.tobytes())
    return len(kmers)


def load_once_kmers(fname, val_bytes=4):
    """(kmer_size, kmers uint64, numseq, numkmer) from a COUNTER_UNIQUE dump"""
    with open(fname, "rb") as f:
        magic, kmer_size, n = struct.unpack("<IBQ", f.read(13))
        if magic != COUNTER_UNIQUE:
            raise IOError("not a dump of unique k-mers")
        if val_bytes == 8:
            rec = np.frombuffer(f.read(), dtype=[("k", "u1"), ("v", "<u8"), ("s", "<u4"), ("p", "<u4")], count=n)
            return kmer_size, rec["v"].astype(np.uint64), rec["s"].copy(), rec["p"].copy()
        rec = np.frombuffer(f.read(), dtype=[("v", "<u4"), ("s", "<u4"), ("p", "<u4")], count=n)
        vals = rec["v"].astype(np.uint64)
        if kmer_size <= 14:
            vals = vals & np.uint64(0x0FFFFFFF)
        return kmer_size, vals, rec["s"].copy(), rec["p"].copy()


def load_kmer_counter(fname, val_bytes):
    """(kmer_size, canonical values uint64, counts uint8) from a COUNTER_MULTIPLE dump"""
    with open(fname, "rb") as f:
        magic, kmer_size, nbc, n = struct.unpack("<IBBQ", f.read(14))
        if magic != COUNTER_MULTIPLE or nbc != 1:
            raise IOError("not a dump of multiple k-mers")
        if val_bytes == 8:
            rec = np.frombuffer(f.read(), dtype=[("k", "u1"), ("v", "<u8"), ("c", "u1")], count=n)
            return kmer_size, rec["v"].astype(np.uint64), rec["c"].copy()
        rec = np.frombuffer(f.read(), dtype=[("v", "<u4"), ("c", "u1")], count=n)
        vals = rec["v"].astype(np.uint64)
        if kmer_size <= 14:
            vals = vals & np.uint64(0x0FFFFFFF)
        return kmer_size, vals, rec["c"].copy()


# ---- sketching parameters as JSON (serde_json::to_writer of the structs) ---------------------------------------------------
_ALGO_NAMES = ["PROB3A", "SUPER", "SUPER2", "OPTDENS", "REVOPTDENS", "HLL"]  # enum SketchAlgo, src/sketcharg.rs:26-33
_DATA_NAMES = ["DNA", "AA"]                                                  # enum DataType, src/sketcharg.rs:13-16


def dump_sketcher_params_json(filename, kmer_size, sketch_size, algo=None, data_t=None):
    """SeqSketcherParams::dump_json (src/sketcharg.rs:79-106): {"kmer_size":..,"sketch_size":..,"algo":"PROB3A","data_t":"DNA"};
    SeqSketcher::dump_json (src/sketching/seqsketchjaccard.rs:142-169) when algo / data_t are None: the two sizes only.
    serde writes unit variants as their names, fields in declaration order, no whitespace."""
    d = {"kmer_size": int(kmer_size), "sketch_size": int(sketch_size)}
    if algo is not None:
        d["algo"] = algo if isinstance(algo, str) else _ALGO_NAMES[algo]
        d["data_t"] = data_t if isinstance(data_t, str) else _DATA_NAMES[data_t or 0]
    with open(filename, "w") as f:
        json.dump(d, f, separators=(",", ":"))


def reload_sketcher_params_json(dirpath):
    """reload_json(dirpath): reads <dirpath>/sketchparams_dump.json (sketcharg.rs:109-138, seqsketchjaccard.rs:172-201)"""
    with open(os.path.join(dirpath, "sketchparams_dump.json")) as f:
        d = json.load(f)
    if "algo" in d and d["algo"] not in _ALGO_NAMES:
        raise ValueError("unknown SketchAlgo %r" % d["algo"])
    return d
