"""Host-side mirror of the reference's read loaders, on top of kmu_ingest_fastx (the parsing runs on the device).

  parse_with_needletail(args) -> Vec<Sequence>      src/io.rs:12-72
  readblockseq(reader, nbseq) -> Vec<Sequence>      src/bin/datasketcher.rs:358-388
A `Sequence` list is represented the way the rest of this package takes sequences: (bases, offsets) arrays.
"""
import numpy as np

from . import lib


def parse_fastq_text(text, ctx=None):
    """All accepted reads (ACGTacgt only) of a FASTQ (or FASTA) text, in file order, plus the counters the reference prints
    (nb rec loaded, nb_bases, nb_bad_bases, nb_bad_read: io.rs:63-68).  `text`: bytes / numpy uint8 (host) or a torch
    uint8 tensor on the device (outputs then stay on the device)."""
    ctx = ctx or lib.Context()
    bases, offsets, info = ctx.ingest_fastx(text)  # FASTQ or FASTA by the first byte, like needletail::parse_fastx_file
    stats = dict(nb_rec_loaded=int(info.n_kept), nb_bases=int(info.n_bases), nb_bad_bases=int(info.nb_bad_bases),
                 nb_bad_read=int(info.nb_bad_reads), nb_records=int(info.n_records))
    return bases, offsets, stats


def parse_with_needletail(filename, ctx=None):
    """parse_with_needletail (src/io.rs:12-72) for an uncompressed FASTQ or FASTA file: the file is read as one byte
    array (np.fromfile) and parsed / filtered / compacted on the device."""
    text = np.fromfile(filename, dtype=np.uint8)
    return parse_fastq_text(text, ctx)


def readblockseq(bases, offsets, first, nbseq):
    """readblockseq (datasketcher.rs:358-388): the next block of at most `nbseq` accepted reads, as a view
    (bases, offsets) of the arrays returned by parse_fastq_text -- the pack-of-10 000 batching of
    datasketcher.rs:243-260 without re-reading the file."""
    n = len(offsets) - 1
    last = min(n, first + nbseq)
    b0, b1 = int(offsets[first]), int(offsets[last])
    return bases[b0:b1], offsets[first:last + 1] - offsets[first]
