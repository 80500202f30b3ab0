"""`datasketcher -f reads.fastq -k 8 -s 200 -d out.sig [-b block_size]` on the GPU path.

Mirror of the reference's tool (src/bin/datasketcher.rs:40-312) without its `ann` sub-command (HNSW is outside the
path): the FASTQ file is parsed and filtered on the device (kmu_ingest_fastq), the accepted reads are sketched with
ProbMinHash3a on canonical `Kmer32bit` k-mers hashed by `int32_hash` (the closure of datasketcher.rs:222-226), whole or
by blocks, and the signatures are written in the reference's dump format (`kmerutils_amd/formats.py`).
"""
import argparse
import sys
import time

import numpy as np

from . import _abi as A
from . import formats, lib


def main(argv=None):
    ap = argparse.ArgumentParser(prog="datasketcher", description=__doc__.splitlines()[0])
    ap.add_argument("-f", "--file", required=True, help="expecting a fastq file")
    ap.add_argument("-s", "--sketch", type=int, required=True, help="expecting sketch size")
    ap.add_argument("-k", "--kmer", type=int, required=True, help="expecting a kmer size (Kmer32bit: <= 14)")
    ap.add_argument("-d", "--dumpfile", required=True, help="expecting name of dumpfile for signature")
    ap.add_argument("-b", "--block_size", type=int, default=0, help="-b for blocksize if sketching by block")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    if not 1 <= args.kmer <= 14:
        ap.error("Kmer32bit holds at most 14 bases (src/base/kmer32bit.rs)")
    ctx = lib.Context(args.device)
    t0 = time.time()
    text = np.fromfile(args.file, dtype=np.uint8)
    bases, offsets, info = ctx.ingest_fastx(text)
    if info.nb_bad_reads:
        print(" number of non acgt sequences %d " % info.nb_bad_reads, file=sys.stderr)  # datasketcher.rs:382-384
    n = int(info.n_kept)
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER32BIT, args.kmer, args.sketch, A.SIG_U32, A.HASHER_NOHASH,
                       A.FHASH_CANON_INVHASH, args.block_size, A.MODE_PER_SEQ, A.INPUT_ASCII, A.MEM_HOST, 0)
    pack = 5000 if args.block_size else 10000  # sequence_pack, datasketcher.rs:212
    if args.block_size:
        out = formats.create_block_signature_dump(args.dumpfile, args.sketch, args.kmer, args.block_size)
    else:
        out = formats.create_signature_dump(args.dumpfile, args.sketch, args.kmer)
    nbseq = 0
    while nbseq < n:
        last = min(n, nbseq + pack)
        b0, b1 = int(offsets[nbseq]), int(offsets[last])
        boff = np.ascontiguousarray(offsets[nbseq:last + 1] - offsets[nbseq])
        if args.block_size:
            bro = ctx.block_layout(boff, args.block_size)
            rows = np.asarray(ctx.sketch(bases[b0:b1], boff, p, block_row_offsets=bro))
            nb = np.diff(bro.astype(np.int64))
            numseq = np.repeat(np.arange(len(nb), dtype=np.uint32) + nbseq, nb)
            numblock = np.concatenate([np.arange(k, dtype=np.uint32) for k in nb]) if len(nb) else np.zeros(0, np.uint32)
            formats.dump_blocks(out, rows, numseq, numblock)
        else:
            formats.dump_signatures_block_u32(np.asarray(ctx.sketch(bases[b0:b1], boff, p)), out)
        nbseq = last
    out.close()
    print(" nb sequences sketched %d, elapsed time (s) %.3f" % (n, time.time() - t0), file=sys.stderr)
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
