"""The count table's 8-byte-per-slot (quotient) format.

Big tables take it by themselves (>= 2^23 slots with 8-bit counters, >= 2^29 with 16-bit ones: the bench's table, and every
count test of the suite whose capacity hint is >= 5.6 M with 8-bit counters); here it is FORCED (`KMU_COUNT_FMT=quot` raises a
small table to the size the format starts at) on the count tests of the other files, so that every reader and writer of a slot
-- direct insertion, the LDS region build on empty and on occupied tables, spill list, query, statistics, dump, export /
merge / retain, eliminate-once, once-positions, the MERGE finalize with its tombstones -- runs on quotient slots against the
same oracle answers (reference contract: src/base/kmercount.rs:241-287, KATs :1524-1617)."""
import importlib.util
import os

import numpy as np
import pytest

from kmerutils_amd import _abi as A
from kmerutils_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location("_quot_" + name, os.path.join(HERE, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def ctx():
    from kmerutils_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


@pytest.fixture
def quot(monkeypatch):
    monkeypatch.setenv("KMU_COUNT_FMT", "quot")
    return monkeypatch


def test_table_formats(ctx, monkeypatch):
    """which table gets which format, and what kmu_count_table_info says"""
    for bits, hint, want in ((8, 1 << 16, 12), (8, 5_600_000, 8), (16, 12_000_000, 12), (16, 400_000_000, 8)):
        c = ctx.counter(A.KMER64BIT, 31, bits, hint)
        ti = c.table_info()
        assert ti["bytes_per_slot"] == want and ti["table_bytes"] == ti["nslots"] * want, (bits, hint, ti)
        assert ti["nslots"] >= 1.5 * hint and ti["nslots"] & (ti["nslots"] - 1) == 0
        if want == 8:  # the count field: the region bits of the table; its ceiling is above what the counter reports
            assert ti["count_field_bits"] == ti["nslots"].bit_length() - 1 - 12
            assert (1 << ti["count_field_bits"]) - 1024 >= (1 << bits) - 1
        c.close()
    monkeypatch.setenv("KMU_COUNT_FMT", "wide")
    c = ctx.counter(A.KMER64BIT, 31, 8, 5_600_000)
    assert c.table_info()["bytes_per_slot"] == 12
    c.close()
    monkeypatch.setenv("KMU_COUNT_FMT", "quot")
    for bits, lg in ((8, 23), (16, 29)):
        c = ctx.counter(A.KMER64BIT, 31, bits, 1024)
        assert c.table_info() == {"nslots": 1 << lg, "table_bytes": 8 << lg, "bytes_per_slot": 8, "count_field_bits": lg - 12,
                                  "count_ceiling": (1 << (lg - 12)) - 1024}  # (2^11 - 1024 = 1 024 for the smallest quotient table)
        c.close()


def test_saturation_of_the_count_field(ctx, oracle, quot):
    """a k-mer seen more often than the count field holds: direct insertion and the LDS region build both stop below the key
    bits (no carry into the stored hash), the reported count is the counter's maximum, and the neighbours in the probe chain
    are still found"""
    bases, off = synth.ont_reads(300, 2_000_000, 0xCA)
    poly = np.frombuffer(b"A" * 150_000 + b"C" * 70_000, np.uint8)  # 149 970 x poly-A 31-mer: > 2^11 - 1024 (8-bit: w = 11) and > 2^17 - 1024
    allb = np.concatenate([bases, poly[:150_000], poly[150_000:]])
    alloff = np.concatenate([off, [off[-1] + 150_000, off[-1] + 220_000]]).astype(np.uint64)
    for bits in (8, 16):
        o = oracle.Counter(A.KMER64BIT, 31, bits, 1 << 22)
        o.add_reads(allb, alloff)
        wk, wc = o.dump(1)
        for path in ("partitioned", "direct"):
            quot.setenv("KMU_COUNT_PATH", path)
            c = ctx.counter(A.KMER64BIT, 31, bits, 1 << 16)
            c.add_reads(allb, alloff)
            gk, gc = c.dump(1)
            assert np.array_equal(gk, wk) and np.array_equal(gc, wc), (bits, path)
            assert gc.max() == (1 << bits) - 1
            assert (c.nb_distinct(), c.nb_unique()) == (o.nb_distinct(), o.nb_unique())
            # an inexact total reads "saturated", not "k-mers lost" (ADVICE r03): the two homopolymers sit at the table's ceiling,
            # the sum of the counts held is short by exactly what they lost
            ti = c.table_info()
            assert ti["count_ceiling"] == (1 << ti["count_field_bits"]) - 1024
            over = [n for n in (149_970, 69_970) if n >= ti["count_ceiling"]]  # (8-bit counters: both; 16-bit, w = 17: poly-A only)
            assert c.nb_saturated() == len(over) >= 1
            lost = sum(n - ti["count_ceiling"] for n in over)
            nk_all = int(np.maximum(np.diff(alloff.astype(np.int64)) - 30, 0).sum())
            assert lost > 0 and c.nb_occurrences() == nk_all - lost
            c.add_reads(allb, alloff)  # onto the saturated fields
            o2 = oracle.Counter(A.KMER64BIT, 31, bits, 1 << 22)
            o2.add_reads(allb, alloff)
            o2.add_reads(allb, alloff)
            assert np.array_equal(c.query(wk), o2.query(wk)), (bits, path)
            c.close()
        quot.delenv("KMU_COUNT_PATH")


PARITY = _load("test_gpu_parity")


@pytest.mark.parametrize("kmer_type,k", [(A.KMER64BIT, 31), (A.KMER16B32BIT, 16), (A.KMER32BIT, 12)])
def test_count_parity_quot(ctx, oracle, quot, kmer_type, k):
    PARITY._count_parity(ctx, oracle, kmer_type, k)


def test_partitioned_two_level_quot(ctx, oracle, quot):
    PARITY.test_count_partitioned_two_level_vs_direct_and_oracle(ctx, oracle)


def test_reference_kat_quot(ctx, oracle, quot):
    PARITY.test_count_reference_kat(ctx, oracle)


def test_add_kmers_partitioned_quot(ctx, oracle, quot):
    PARITY.test_count_add_kmers_partitioned(ctx, oracle)


def test_extract_by_owner_quot(ctx, oracle, quot):
    PARITY.test_extract_by_owner_single_gpu(ctx, oracle)


def test_single_pass_spill_list_quot(ctx, oracle, quot):
    PARITY.test_count_single_pass_spill_list(ctx, oracle, quot)


@pytest.mark.parametrize("pct", ["100", "60"])
def test_single_pass_partition_quot(ctx, oracle, quot, pct):
    PARITY.test_count_single_pass_partition(ctx, oracle, quot, pct)


def test_once_kmers_quot(ctx, oracle, quot, tmp_path):
    PARITY.test_once_kmers_with_positions(ctx, oracle, tmp_path)


@pytest.mark.parametrize("owner", ["minimizer", "hash"])
def test_distributed_counter_quot(oracle, quot, owner):
    """both routes of a distributed add through RCCL at world size 1, incl. the MERGE finalize (owner census, emit, zeroed
    counts as tombstones) on quotient slots, with either owner function"""
    _load("test_gpu_comm").test_rccl_world1_distributed_counter(oracle, quot, owner)


def test_sketch_count_quot(ctx, oracle, quot):
    PIPE = _load("test_gpu_pipeline")
    PIPE.test_sketch_count_host_and_device(ctx, oracle, quot, "1", "1")
    PIPE.test_sketch_count_host_chunked_level1(ctx, oracle, quot)


@pytest.mark.parametrize("seed", range(6))
def test_counter_usage_sweep_quot(ctx, oracle, quot, seed):
    FUZZ = _load("test_gpu_fuzz")
    FUZZ.test_counter_usage_sweep(ctx, oracle, seed, quot)
    FUZZ.test_count_sweep(ctx, oracle, seed)


@pytest.mark.parametrize("genome,n_reads,what", [(4_000_000, 4000, "ratio ~1"), (150_000, 4000, "ratio ~3.6"), (10_000, 4000, "ratio ~50")])
def test_table_sized_from_the_measured_duplication(oracle, genome, n_reads, what):
    """KMU_COUNT_HINT_OCCURRENCES: the hint counts k-mer occurrences, the first add measures occurrences / distinct on a key
    sample of its batch and allocates 1.5 x occurrences / ratio slots (config 4's shard: 2^29 instead of 2^31).  The reference
    sizes its filters blind (capacity 3e9 / n, src/base/kmercount.rs:888-892); counts are the same whatever the size."""
    import torch
    from kmerutils_amd import lib
    bases, off = synth.genome_reads(n_reads, np.full(n_reads, 150, np.int64), genome, 0xD1, sub=0.002)
    nk = n_reads * 120
    g = oracle.Counter(A.KMER64BIT, 31, 16, 1 << 20)
    g.add_reads(bases, off)
    wk, wc = g.dump(1)
    ctx = lib.Context(0)
    blind = ctx.counter(A.KMER64BIT, 31, 16, nk)
    blind_slots = blind.table_info()["nslots"]
    blind.close()
    for dev in (True, False):
        c = ctx.counter(A.KMER64BIT, 31, 16, nk, hint_occurrences=True)
        assert c.table_info()["nslots"] == 0  # nothing allocated yet
        if dev:
            c.add_reads(torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda())
        else:
            c.add_reads(bases, off)
        slots = c.table_info()["nslots"]
        gk, gc = c.dump(1)
        assert np.array_equal(gk, wk) and np.array_equal(gc, wc), what
        # load <= 2/3, and no more than the power of two that 1.5 x 1.1 x distinct asks for
        assert 1.5 * wk.size <= slots <= max(1024, 2 * 1.5 * 1.1 * wk.size + 2048), (what, slots, wk.size, nk / wk.size)
        assert slots <= blind_slots
        # a second batch goes into the table the first one made
        c.add_reads(bases, off)
        assert c.table_info()["nslots"] == slots
        assert np.array_equal(c.dump(1)[1], 2 * wc)
        c.close()
    # readers before the first add see an empty table of the hinted size; explicit k-mers size it by their number
    c = ctx.counter(A.KMER64BIT, 31, 16, nk, hint_occurrences=True)
    assert c.nb_distinct() == 0 and (c.query(wk[:10].copy()) == 0).all()
    c.close()
    c = ctx.counter(A.KMER64BIT, 31, 16, 16, hint_occurrences=True)
    c.add_kmers(wk)
    assert c.table_info()["nslots"] >= 1.5 * wk.size and c.nb_distinct() == wk.size
    c.close()
    ctx.close()


@pytest.mark.parametrize("leaf6", ["1", "0"])
def test_six_byte_leaf_items_of_big_tables(ctx, oracle, monkeypatch, leaf6):
    """tables of >= 2^28 slots (the region index takes w >= 16 hash bits): level 2 of the partitioned build leaves 6 bytes per item
    in two planes, the region build reads them back (KMU_COUNT_LEAF6; round 4).  A 2 GB table with a batch the oracle can count,
    through the reads path, the k-mer array path and the super-k-mer records path."""
    import torch
    monkeypatch.setenv("KMU_COUNT_LEAF6", leaf6)
    monkeypatch.setenv("KMU_COUNT_PATH", "partitioned")
    monkeypatch.setenv("KMU_COUNT_SEG", "2")
    bases, off = synth.ont_reads(700, 500_000, 0xCB)
    poly = np.frombuffer(b"A" * 3000 + b"ACGTTGCA" * 400, np.uint8)
    allb = np.concatenate([bases, poly])
    alloff = np.concatenate([off, [off[-1] + 3000, off[-1] + 6200]]).astype(np.uint64)
    o = oracle.Counter(A.KMER64BIT, 31, 16, 1 << 22)
    o.add_reads(allb, alloff)
    wk, wc = o.dump(1)
    hint = 170_000_000  # x 1.5 -> 2^28 slots of 8 bytes, w = 16
    c = ctx.counter(A.KMER64BIT, 31, 8, hint)
    ti = c.table_info()
    assert ti["nslots"] == 1 << 28 and ti["bytes_per_slot"] == 8 and ti["count_field_bits"] == 16
    c.add_reads(torch.from_numpy(allb).cuda(), torch.from_numpy(alloff.astype(np.int64)).cuda())
    gk, gc = c.dump(1)
    assert np.array_equal(gk, wk) and np.array_equal(gc, np.minimum(wc, 255))
    c.add_reads(allb, alloff)  # a second batch into the occupied table (the build reads the image first)
    assert np.array_equal(c.query(wk), np.minimum(2 * wc.astype(np.int64), 255))
    c.close()
    c = ctx.counter(A.KMER64BIT, 31, 8, hint)  # explicit k-mers: the array path's two levels
    c.add_kmers(torch.from_numpy(np.repeat(wk, 2).view(np.int64)).cuda())
    assert np.array_equal(c.dump(1)[0], wk) and (c.dump(1)[1] == 2).all()
    c.close()
    src = ctx.counter(A.KMER64BIT, 31, 8, 1 << 16)  # super-k-mer records: the receiver's level 1
    rec, bounds, kmers = src.extract_superkmers(allb, alloff, 3)
    ctx.synchronize()
    c = ctx.counter(A.KMER64BIT, 31, 8, hint)
    c.add_superkmers(rec)
    gk, gc = c.dump(1)
    assert np.array_equal(gk, wk) and np.array_equal(gc, np.minimum(wc, 255))
    c.close()
    src.close()
