"""The count table's 8-byte-per-slot (quotient) format.

Big tables take it by themselves (more than 2048 regions with 8-bit counters, >= 2^17 with 16-bit ones: the bench's table, and
every count test of the suite whose capacity hint is >= 5.9 M with 8-bit counters); here it is FORCED (`KMU_COUNT_FMT=quot` raises a
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
    """which table gets which format, and what kmu_count_table_info says: whole regions of 4096 slots at a load of 0.70 at the
    hint (not a power of two), 8-byte slots where the region index is worth the count field"""
    for bits, hint, want in ((8, 1 << 16, 12), (8, 5_600_000, 12), (8, 6_000_000, 8), (16, 12_000_000, 12), (16, 400_000_000, 8)):
        c = ctx.counter(A.KMER64BIT, 31, bits, hint)
        ti = c.table_info()
        assert ti["bytes_per_slot"] == want and ti["table_bytes"] == ti["nslots"] * want, (bits, hint, ti)
        assert hint / 0.70 <= ti["nslots"] <= hint / 0.70 * 1.11 + 4096 and ti["nslots"] % 4096 == 0, (bits, hint, ti)
        if want == 8:  # the count field: the bits the region index saves; its ceiling is above what the counter reports
            n_regions = ti["nslots"] // 4096
            assert (1 << ti["count_field_bits"]) <= n_regions < (4 << ti["count_field_bits"])
            assert (1 << ti["count_field_bits"]) - 1024 >= (1 << bits) - 1
        c.close()
    monkeypatch.setenv("KMU_COUNT_FMT", "wide")
    c = ctx.counter(A.KMER64BIT, 31, 8, 6_000_000)
    assert c.table_info()["bytes_per_slot"] == 12
    c.close()
    monkeypatch.setenv("KMU_COUNT_FMT", "quot")
    for bits, lg in ((8, 23), (16, 29)):
        c = ctx.counter(A.KMER64BIT, 31, bits, 1024)
        assert c.table_info() == {"nslots": 1 << lg, "table_bytes": 8 << lg, "bytes_per_slot": 8, "count_field_bits": lg - 12,
                                  "count_ceiling": (1 << (lg - 12)) - 1024}  # (2^11 - 1024 = 1 024 for the smallest quotient table)
        c.close()


# region maps that are not powers of two (round 5): (KMU_COUNT_REGIONS, what it exercises)
REGION_MAPS = [("1", "one region"), ("3", "three regions, one level"), ("1021", "a prime, one level"), ("2048", "the largest single level"),
               ("6144", "3 x 2^11: 128 groups of 48"), ("2052", "just over one level: 64 groups of 36"), ("33000", "256 groups of 132"),
               ("66000", "512 groups of 132: 6-byte leaves"), ("300000", "1024 groups of 296: 9.9 GB")]


@pytest.mark.parametrize("regions,what", REGION_MAPS, ids=[r for r, _ in REGION_MAPS])
def test_region_maps_that_are_no_powers_of_two(ctx, oracle, monkeypatch, regions, what):
    """KMU_COUNT_REGIONS forces the table's region count whatever the hint: every way into the table and out of it on maps of 1 .. 3e5
    regions -- single-pass partition, exact levels, direct insertion, a second batch onto the occupied image, query, dump,
    statistics, export_part / merge, saturation of the count field, the spill list of overflowing streams -- gives the oracle's
    counts (kmercount.rs:241-287).  Tables of >= 2048 regions are quotient tables here (8-bit counters: w >= 11)."""
    import torch
    monkeypatch.setenv("KMU_COUNT_REGIONS", regions)
    n_regions = int(regions)
    # (a table of few regions only holds so much: reads for ~1/3 of its slots, at most ~1.2 M k-mers for the oracle)
    n_kmers = min(1_200_000, max(600, n_regions * 4096 // 3))
    n_reads = max(2, n_kmers // 3000)
    bases, off = synth.genome_reads(n_reads, np.full(n_reads, n_kmers // n_reads + 30, np.int64), 4_000_000, 0xD7, sub=0.01)
    poly = np.frombuffer(b"A" * 2500, np.uint8)  # 2 470 x poly-A: beyond 2^11 - 1024 of the smallest quotient tables, and beyond 255
    if n_regions > 1:
        bases = np.concatenate([bases, poly])
        off = np.concatenate([off, [off[-1] + poly.size]]).astype(np.uint64)
    o = oracle.Counter(A.KMER64BIT, 31, 16, 1 << 22)
    o.add_reads(bases, off)
    wk, wc = o.dump(1)
    want8 = np.minimum(wc, 255)
    d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
    for path, seg, pct in (("partitioned", "2", "100"), ("partitioned", "2", "60"), ("partitioned", "0", "100"), ("direct", "1", "100")):
        monkeypatch.setenv("KMU_COUNT_PATH", path)
        monkeypatch.setenv("KMU_COUNT_SEG", seg)
        monkeypatch.setenv("KMU_COUNT_SEG_PCT", pct)
        c = ctx.counter(A.KMER64BIT, 31, 8, 1024)
        ti = c.table_info()
        rounded = ti["nslots"] // 4096
        assert n_regions <= rounded <= n_regions + 4 * 2048 and (rounded == n_regions if n_regions <= 2048 else rounded % 4 == 0), (ti, what)
        assert ti["bytes_per_slot"] == (8 if n_regions >= 2048 else 12), (ti, what)
        c.add_reads(d_b, d_o)
        gk, gc = c.dump(1)
        assert np.array_equal(gk, wk) and np.array_equal(gc, want8), (what, path, seg, pct)
        assert (c.nb_distinct(), c.nb_unique()) == (o.nb_distinct(), o.nb_unique())
        c.add_reads(bases, off)  # a second batch: onto the occupied image (the build reads the regions first) / direct insertion
        assert np.array_equal(c.query(wk), np.minimum(2 * wc.astype(np.int64), 255)), (what, path, seg, pct)
        # export by key partition and merge into a table with ANOTHER map: the keys come back out of the quotients
        monkeypatch.setenv("KMU_COUNT_REGIONS", "5000")  # (128 groups of 40)
        c2 = ctx.counter(A.KMER64BIT, 31, 8, 1024)
        monkeypatch.setenv("KMU_COUNT_REGIONS", regions)
        tot = 0
        for part in range(3):
            kk, cc = c.export_part(part, 3)
            tot += kk.size
            c2.merge_entries(kk, cc)
        assert tot == wk.size and np.array_equal(c2.dump(1)[0], wk)
        ceil = ti["count_ceiling"] if ti["bytes_per_slot"] == 8 else 1 << 32
        assert np.array_equal(c2.query(wk), np.minimum(np.minimum(2 * wc.astype(np.int64), ceil), 255))
        c2.close()
        c.close()
    # explicit k-mers (the array path's levels) into the same map
    monkeypatch.setenv("KMU_COUNT_PATH", "partitioned")
    monkeypatch.setenv("KMU_COUNT_SEG", "2")
    c = ctx.counter(A.KMER64BIT, 31, 8, 1024)
    c.add_kmers(torch.from_numpy(np.repeat(wk, 3).view(np.int64)).cuda())
    gk, gc = c.dump(1)
    assert np.array_equal(gk, wk) and (gc == 3).all()
    c.close()


def test_table_load_factor_switch(ctx, oracle, monkeypatch):
    """KMU_COUNT_LOAD (per cent; A/B runs of the region build): the same counts from a table at a load of 0.85 at the hint"""
    bases, off = synth.ont_reads(400, 1_500_000, 0xD8)
    o = oracle.Counter(A.KMER64BIT, 31, 8, 1 << 22)
    o.add_reads(bases, off)
    wk, wc = o.dump(1)
    monkeypatch.setenv("KMU_COUNT_PATH", "partitioned")
    o.add_reads(bases, off)
    wk2, wc2 = o.dump(1)
    for load in ("85", "40", "90"):
        monkeypatch.setenv("KMU_COUNT_LOAD", load)
        c = ctx.counter(A.KMER64BIT, 31, 8, wk.size)
        assert abs(c.table_info()["nslots"] - wk.size / (int(load) / 100)) <= 4096
        c.add_reads(bases, off)
        gk, gc = c.dump(1)
        assert np.array_equal(gk, wk) and np.array_equal(gc, wc)
        # the same reads onto the full table: the region build starts from the slab, every item walks to its key -- at a load of 0.9
        # more items of a wave are left after the batched probes than its pool holds (k_part_build_q: the in-lane walk)
        c.add_reads(bases, off)
        gk, gc = c.dump(1)
        assert np.array_equal(gk, wk2) and np.array_equal(gc, wc2), load
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
    sample of its batch and allocates occurrences / ratio / 0.70 slots (config 4's shard: a quarter of what the occurrences ask for).  The reference
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
        # a load of <= 0.70, and no more than the sample's error and the rounding to whole regions add
        assert wk.size / 0.70 <= slots <= max(1024, (1.16 * wk.size + 1024) / 0.70 + 4096 * 260), (what, slots, wk.size, nk / wk.size)
        assert slots <= max(blind_slots, int(off[-1]) / 0.70 + 2 * 4096)  # (no duplication to speak of: the bases of the batch bound its k-mers)
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
    assert c.table_info()["nslots"] >= wk.size / 0.70 and c.nb_distinct() == wk.size
    c.close()
    ctx.close()


@pytest.mark.parametrize("leaf6", ["1", "0"])
def test_six_byte_leaf_items_of_big_tables(ctx, oracle, monkeypatch, leaf6):
    """tables whose region index is worth w >= 16 hash bits: level 2 of the partitioned build leaves the <= 48 bits a slot keeps of
    an item (6 bytes, blocks of eight), the region build reads them back (KMU_COUNT_LEAF6; round 4).  A 2.3 GB table -- 512 groups of
    140 regions: no power of two -- with a batch the oracle can count,
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
    hint = 200_000_000  # / 0.70 -> 69 755 regions -> 512 groups of 140: slots of 8 bytes, w = 9 + 7 = 16
    c = ctx.counter(A.KMER64BIT, 31, 8, hint)
    ti = c.table_info()
    assert ti["nslots"] == 512 * 140 * 4096 and ti["bytes_per_slot"] == 8 and ti["count_field_bits"] == 16
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
