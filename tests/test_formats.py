"""On-disk dump formats of the reference (host side): literal header bytes worked out from the writers' source, and
round trips through the readers."""
import struct

import numpy as np
import pytest

from kmerutils_amd import formats as F


def test_signature_dump_layout_and_roundtrip(tmp_path):
    fn = str(tmp_path / "sig.bin")
    sigs = np.arange(3 * 5, dtype=np.uint32).reshape(3, 5) * 7919
    f = F.create_signature_dump(fn, 5, 16)
    F.dump_signatures_block_u32(sigs, f)
    f.close()
    raw = open(fn, "rb").read()
    # seqsketchjaccard.rs:404-411: magic, sig_size = 4, sketch_size, kmer_size, little-endian u32 each
    assert raw[:16] == bytes.fromhex("ddeaabce") + struct.pack("<III", 4, 5, 16)
    assert raw[16:20] == struct.pack("<I", 0) and raw[20:24] == struct.pack("<I", 7919) and len(raw) == 16 + 60
    r = F.SigSketchFileReader(fn)
    assert (r.get_kmer_size(), r.get_signature_length(), r.get_signature_size()) == (16, 5, 4)
    assert np.array_equal(r.next(), sigs[0]) and np.array_equal(r.next(), sigs[1]) and np.array_equal(r.next(), sigs[2])
    assert r.next() is None
    with pytest.raises(ValueError):
        F.dump_signatures_block_u32(sigs.astype(np.uint64), open(fn, "wb"))
    open(fn, "wb").write(b"\x00" * 16)
    with pytest.raises(IOError):
        F.SigSketchFileReader(fn)


def test_block_dump_layout_and_roundtrip(tmp_path):
    fn = str(tmp_path / "blk.bin")
    rows = (np.arange(5 * 4, dtype=np.uint32).reshape(5, 4) + 1) * 3
    numseq = np.array([10, 10, 10, 11, 11], np.uint32)
    numblock = np.array([0, 1, 2, 0, 1], np.uint32)
    f = F.create_block_signature_dump(fn, 4, 12, 1000)
    F.dump_blocks(f, rows, numseq, numblock)
    f.close()
    raw = open(fn, "rb").read()
    # seqblocksketch.rs:215-223: 17 bytes -- magic u32, sig_size u8, sketch_size, kmer_size, block_size u32
    assert raw[:17] == bytes.fromhex("ddbaabce") + b"\x04" + struct.pack("<III", 4, 12, 1000)
    # seqblocksketch.rs:175-188 + :59-65: numseq, nbblock, then per block numseq, numblock, sketch
    assert raw[17:25] == struct.pack("<II", 10, 3) and raw[25:33] == struct.pack("<II", 10, 0)
    assert raw[33:49] == rows[0].tobytes()
    assert len(raw) == 17 + 2 * 8 + 5 * (8 + 16)
    r = F.SigBlockSketchFileReader(fn)
    assert (r.sketch_size, r.kmer_size, r.block_size) == (4, 12, 1000)
    ns, blocks = r.next()
    assert ns == 10 and [b[0] for b in blocks] == [0, 1, 2] and np.array_equal(blocks[2][1], rows[2])
    ns, blocks = r.next()
    assert ns == 11 and len(blocks) == 2 and np.array_equal(blocks[1][1], rows[4])
    assert r.next() is None


def test_kmer_count_dump_layout_and_roundtrip(tmp_path):
    fn = str(tmp_path / "cnt.bin")
    kmers = np.array([5, 77, 1 << 40], np.uint64)
    counts = np.array([2, 300, 9], np.uint32)
    assert F.dump_kmer_counter(fn, kmers, counts, 31, 8) == 3
    raw = open(fn, "rb").read()
    # kmercount.rs:487-500: magic u32, kmer_size u8, nb_bytes_by_count u8, nb u64; kmer64bit.rs:98-104: size byte + u64; count u8
    assert raw[:14] == bytes.fromhex("ffbba2ce") + bytes([31, 1]) + struct.pack("<Q", 3)
    assert raw[14:24] == bytes([31]) + struct.pack("<Q", 5) + bytes([2]) and len(raw) == 14 + 3 * 10
    k, v, c = F.load_kmer_counter(fn, 8)
    assert k == 31 and np.array_equal(v, kmers) and c.tolist() == [2, 255, 9]
    # Kmer32bit: the dumped word carries k in its top nibble (kmer32bit.rs:141-144)
    F.dump_kmer_counter(fn, np.array([0x2A], np.uint64), np.array([4]), 8, 4)
    raw = open(fn, "rb").read()
    assert raw[14:19] == struct.pack("<I", 0x2A | (8 << 28)) + bytes([4])
    k, v, c = F.load_kmer_counter(fn, 4)
    assert k == 8 and v.tolist() == [0x2A] and c.tolist() == [4]


def test_sketcher_params_json(tmp_path):
    """serde_json layout of SeqSketcherParams / SeqSketcher (src/sketcharg.rs:40-138, seqsketchjaccard.rs:117-201)"""
    fn = tmp_path / "sketchparams_dump.json"
    F.dump_sketcher_params_json(str(fn), 8, 200, "PROB3A", "DNA")
    assert fn.read_text() == '{"kmer_size":8,"sketch_size":200,"algo":"PROB3A","data_t":"DNA"}'
    assert F.reload_sketcher_params_json(str(tmp_path)) == dict(kmer_size=8, sketch_size=200, algo="PROB3A", data_t="DNA")
    F.dump_sketcher_params_json(str(fn), 12, 128, 4, 1)  # enum ordinals: REVOPTDENS, AA
    assert fn.read_text() == '{"kmer_size":12,"sketch_size":128,"algo":"REVOPTDENS","data_t":"AA"}'
    F.dump_sketcher_params_json(str(fn), 16, 50)       # SeqSketcher: the two sizes
    assert fn.read_text() == '{"kmer_size":16,"sketch_size":50}'
    assert F.reload_sketcher_params_json(str(tmp_path)) == dict(kmer_size=16, sketch_size=50)
