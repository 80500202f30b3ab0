#!/usr/bin/env python3
"""Stress of the batched region build (k_part_build_q) at sizes the oracle does not reach: the partitioned route against direct
insertion (two GPU paths with nothing in common behind the k-mer walk) on the same reads -- tables at several load factors, a second
add onto the full table (the build starts from the slab: long probe sequences, the per-wave pool overflows), reads with heavy
duplication (count fields near their ceiling), 8- and 16-bit counters.  usage (GPU box): python scripts/r05_build_stress.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmerutils_amd import lib, synth, _abi as A

def run(ctx, bases, off, bits, hint, env):
    for k, v in env.items(): os.environ[k] = v
    try:
        c = ctx.counter(A.KMER64BIT, 31, bits, hint)
        c.add_reads(bases, off)
        d1 = c.dump(1)
        c.add_reads(bases, off)
        d2 = c.dump(1)
        info = c.table_info()
        c.close()
        return d1, d2, info
    finally:
        for k in env: os.environ.pop(k, None)

def main():
    ctx = lib.Context()
    cases = [("ont 4 k reads, cov 5", synth.ont_reads(4000, 5_000_000, 1)),
             ("ont 20 k reads, cov 1", synth.ont_reads(20000, 100_000_000, 2)),
             ("150 bp, 200 kb genome x 300", synth.illumina_reads(400_000, 200_000, 7))]  # (8-bit counters saturate)
    bad = 0
    for name, (bases, off) in cases:
        nk = int(np.maximum(np.diff(off.astype(np.int64)) - 30, 0).sum())
        for bits in (8, 16):
            for load in ("55", "70", "88"):
                t0 = time.time()
                env = {"KMU_COUNT_LOAD": load}
                p1, p2, info = run(ctx, bases, off, bits, nk, dict(env, KMU_COUNT_PATH="partitioned"))
                q1, q2, _ = run(ctx, bases, off, bits, nk, dict(env, KMU_COUNT_PATH="direct"))
                ok = all(np.array_equal(a, b) for a, b in zip(p1 + p2, q1 + q2))
                bad += not ok
                print("%-22s bits %2d load %s: %9d distinct, max count %5d, slots %.2e fmt w=%s  %s  (%.1f s)" % (
                    name, bits, load, p1[0].size, int(p2[1].max()), info["nslots"], info.get("count_field_bits"), "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    print("mismatches:", bad)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
