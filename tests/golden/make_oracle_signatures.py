#!/usr/bin/env python3
"""Writes tests/golden/oracle_signatures.json: the ORACLE's signatures on the reference's own test strings with the
reference's test parameters (k = 5 / m = 4000 is cut to m = 200 here; k = 16 / m = 50, 100; AA k = 5 / m = 400).

These are not reference outputs -- the sketchers' inner arithmetic lives in the un-vendored `probminhash` crate, see the
"parity unpinned" note in oracle/kmu_oracle.h -- they freeze the restatement as it stands, so that a later change to the
oracle (or to a kernel that the oracle would silently follow) shows up as a diff of this file.  If a machine with cargo
ever produces the real crate's signatures for the same inputs, they go into the same slots.

    python tests/golden/make_oracle_signatures.py        # rewrites the fixture
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kmerutils_amd import _abi as A  # noqa: E402
from oracle import oracle as O  # noqa: E402

SEQSTR = b"TCAAAGGGAAACATTCAAAATCAGTATGCGCCCGTTCAGTTACGTATTGCTCTCGCTAATGAGATGGGCTGGGTACAGAG"  # seqsketchjaccard.rs:751
AA1 = b"MTEQIELIKLYSTRILALAAQMPHVGSLDNPDASAMKRSPLCGSKVTVDVIMQNGKITFDGFEVLAPASEYKNRHASILLSLDATAEACASIAAQNSA"  # aautils/setsketchert.rs:1223
AA2 = b"MTEQIELIKLYSTRILALAAQMPHVGSLDNPDASAMKRSPLCGSKVMTEQIELIKLYSTRILALAAQMPHVGSLDNPDASAMKRSPLCGSKV"


def revcomp(s):
    return bytes({65: 84, 67: 71, 71: 67, 84: 65}[c] for c in reversed(s))


DNA = [SEQSTR, SEQSTR[:40], revcomp(SEQSTR)]
DENS = [b"ATCATGCCCCTTTAGAAAATTTCCGGATCATCGTACGGAGCATGCGTACAACGTCGATGC",   # setsketchert.rs:1081-1083
        b"ATCATGCCCCTTTAGAAAATTTCCGGATCATCATGCCCCTTTAGAAAATTTCCGGATC"]
CASES = [
    # name, sequences, algo, kmer_type, k, m, sig, hasher, fhash, flags
    ("pminhasha_kmer_smallb_revcomp", DNA, A.ALGO_PROB3A, A.KMER32BIT, 5, 200, A.SIG_U32, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0),
    ("pminhasha_kmer_smallb_identity", DNA, A.ALGO_PROB3A, A.KMER32BIT, 5, 200, A.SIG_U32, A.HASHER_NOHASH, A.FHASH_IDENTITY_RAW, 0),
    ("pminhasha_k16b32bit", DNA, A.ALGO_PROB3A, A.KMER16B32BIT, 16, 50, A.SIG_U32, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0),
    ("pminhash_kmer64bit", DNA, A.ALGO_PROB3A, A.KMER64BIT, 16, 50, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0),
    ("pminhash_kmer64bit_rand08", DNA, A.ALGO_PROB3A, A.KMER64BIT, 16, 50, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, A.FLAG_RAND08),
    ("superminhash_16b32bit_fnv_f64", DNA, A.ALGO_SUPER, A.KMER16B32BIT, 16, 100, A.SIG_F64, A.HASHER_FNV1A, A.FHASH_CANON_INVHASH, 0),
    ("superminhash_16b32bit_nohash_f32", DNA, A.ALGO_SUPER, A.KMER16B32BIT, 16, 100, A.SIG_F32, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0),
    ("superminhash2_32bit_u64", DNA, A.ALGO_SUPER2, A.KMER32BIT, 12, 64, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0),
    ("seqaa_probminhash_64bit", [AA1, AA2], A.ALGO_PROB3A, A.KMERAA64BIT, 5, 400, A.SIG_U64, A.HASHER_NOHASH, A.FHASH_VALUE_MASKED, 0),
    ("seqaa_probminhash_32bit", [AA1, AA2], A.ALGO_PROB3A, A.KMERAA32BIT, 5, 400, A.SIG_U32, A.HASHER_NOHASH, A.FHASH_VALUE_MASKED, 0),
    ("seqaa_superminhash_64bit_f64", [AA1, AA2], A.ALGO_SUPER, A.KMERAA64BIT, 5, 128, A.SIG_F64, A.HASHER_NOHASH, A.FHASH_VALUE_MASKED, 0),
    # setsketchert.rs:1075-1230 (m cut from 800 / 8000 to 400 / 1000), aautils/setsketchert.rs:1394
    ("seq_optdensminhash_f64", DENS, A.ALGO_OPTDENS, A.KMER32BIT, 5, 400, A.SIG_F64, A.HASHER_NOHASH, A.FHASH_VALUE_MASKED, 0),
    ("seq_revoptdensminhash_f32", DENS, A.ALGO_REVOPTDENS, A.KMER32BIT, 5, 1000, A.SIG_F32, A.HASHER_NOHASH, A.FHASH_VALUE_MASKED, 0),
    ("seqaa_optdensminhash_32bit_f64", [AA1, AA2], A.ALGO_OPTDENS, A.KMERAA32BIT, 5, 80, A.SIG_F64, A.HASHER_NOHASH, A.FHASH_VALUE_MASKED, 0),
    ("hyperloglog_u16_default_params", DNA, A.ALGO_HLL, A.KMER64BIT, 16, 256, A.SIG_U16, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0),
]


def compute(case):
    name, seqs, algo, kt, k, m, sig, hasher, fhash, flags = case
    bases, off = O.concat(seqs)
    p = A.SketchParams(algo, kt, k, m, sig, hasher, fhash, 0, A.MODE_PER_SEQ, A.INPUT_ASCII, A.MEM_HOST, flags)
    rows = O.sketch(bases, off, p)
    if rows.dtype.kind == "f":  # exact: the bit patterns
        rows = rows.view(np.uint32 if rows.dtype.itemsize == 4 else np.uint64)
    return p, bases, off, rows


def main():
    out = {"note": "oracle outputs (restatement, parity unpinned): see make_oracle_signatures.py", "cases": {}}
    for case in CASES:
        name, seqs = case[0], case[1]
        p, _, _, rows = compute(case)
        out["cases"][name] = {"sequences": [s.decode() for s in seqs], "algo": case[2], "kmer_type": case[3], "k": case[4],
                              "m": case[5], "sig_type": case[6], "hasher": case[7], "fhash": case[8], "flags": case[9],
                              "float_rows_as_bits": case[6] in (A.SIG_F32, A.SIG_F64),
                              "rows": [[int(x) for x in r] for r in rows]}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_signatures.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")


if __name__ == "__main__":
    main()
