"""`parsefastq -f reads.fastq kmer --count -s 31` on the GPU path.

Mirror of the counting branch of the reference's tool (src/bin/parsefastq.rs:205-236): the FASTQ file is parsed and
filtered on the device, every canonical k-mer of the accepted reads is counted (`count_kmer_threaded_one_to_many`,
src/base/kmercount.rs:881-974) and the k-mers seen at least twice are written to `<file>.multi_kmer.bin` in the
reference's dump format (`threaded_dump_kmer_counter` / `dump_kmer_counter`, kmercount.rs:467-531, :653-791).  The
k-mer type follows the tool: Kmer64bit for 16 < k <= 32 (here: <= 31, k = 32 is broken upstream), Kmer16b32bit for
k == 16, Kmer32bit for k <= 14.
"""
import argparse
import os
import sys
import time

import numpy as np

from . import _abi as A
from . import formats, lib


def kmer_type_for(k):
    if 16 < k <= 31:
        return A.KMER64BIT, 8
    if k == 16:
        return A.KMER16B32BIT, 4
    if 1 <= k <= 14:
        return A.KMER32BIT, 4
    raise ValueError("no k-mer type for k = %d (parsefastq.rs:215-236)" % k)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="parsefastq", description=__doc__.splitlines()[0])
    ap.add_argument("-f", "--file", required=True)
    ap.add_argument("-s", "--kmer_size", type=int, default=16)
    ap.add_argument("--outdir", default=".", help="directory of <file>.multi_kmer.bin (the tool writes to the cwd)")
    ap.add_argument("--unique", action="store_true", help="dump the 16-mers seen exactly once with their positions "
                    "(KmerProcessing::Unicity, parsefastq.rs:238-247) instead of counting")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    kmer_type, val_bytes = kmer_type_for(args.kmer_size)
    ctx = lib.Context(args.device)
    t0 = time.time()
    text = np.fromfile(args.file, dtype=np.uint8)
    bases, offsets, info = ctx.ingest_fastx(text)
    print(" nb rec loaded = %d \nnb_bases %d\nnb_bad_bases %d\nnb_bad_read %d" %
          (info.n_kept, info.n_bases, info.nb_bad_bases, info.nb_bad_reads), file=sys.stderr)  # io.rs:63-68
    if args.unique:  # filter1_kmer_16b32bit + dump_in_file_once_kmer16b32bit -> <file>.once_kmer.bin
        from . import kmercount
        flt = kmercount.KmerFilter1(16, max(1024, int(info.kept_bases)), ctx=ctx)
        if info.n_kept:
            flt.insert_reads((bases, offsets))
        out = os.path.join(args.outdir, os.path.basename(args.file) + ".once_kmer.bin")
        print("dumping unique kmers in file : %s " % out, file=sys.stderr)
        n = flt.dump_in_file_once_kmer16b32bit(out, (bases, offsets)) if info.n_kept else formats.dump_once_kmers(out, [], [], [], 16)
        print("dump_in_file_once_kmer16b32bit, number of kmer dumped : %d " % n, file=sys.stderr)
        ctx.close()
        return 0
    nk = int(np.maximum(np.diff(offsets.astype(np.int64)) - args.kmer_size + 1, 0).sum())
    counter = ctx.counter(kmer_type, args.kmer_size, 8, max(nk, 1024))
    if info.n_kept:
        counter.add_reads(bases, offsets)
    kmers, counts = counter.dump(2)
    out = os.path.join(args.outdir, os.path.basename(args.file) + ".multi_kmer.bin")  # parsefastq.rs:207-211
    print("dumping multiple kmers in file : %s " % out, file=sys.stderr)
    n = formats.dump_kmer_counter(out, kmers, counts, args.kmer_size, val_bytes)
    print("dump_kmer_counter, number of kmer dumped : %d (distinct %d, unique %d), elapsed time (s) %.3f" %
          (n, counter.nb_distinct(), counter.nb_unique(), time.time() - t0), file=sys.stderr)
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
