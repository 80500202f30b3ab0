"""The C++ host mirror (include/kmerutils.hpp): its test program (tests/cpp/test_mirror.cpp, the reference's own tests
re-stated against the mirror + bit parity with the oracle) and the two tools, run as child processes."""
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

from kmerutils_amd import _abi as A
from kmerutils_amd import build as kbuild
from kmerutils_amd import formats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "cpp"))
import build_mirror as cppbuild  # noqa: E402  (tests/cpp/build_mirror.py)

TEST_BIN = cppbuild.TEST_BIN
DATASKETCHER = os.path.join(ROOT, "kmerutils_amd", "bin", "datasketcher")
PARSEFASTQ = os.path.join(ROOT, "kmerutils_amd", "bin", "parsefastq")

with open(os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp")) as _f:
    CASES = re.findall(r"^TEST\((\w+)\)", _f.read(), flags=re.M)


@pytest.fixture(scope="module")
def host_programs():
    kbuild.build_host()
    return cppbuild.build()


def test_host_programs_build_and_refuse_to_run_without_a_device(host_programs):
    """g++ builds the mirror's programs against libkmu.so; without a GPU they stop with the library's error, they do
    not compute anything on the CPU"""
    import torch
    assert len(CASES) >= 15
    for exe in (TEST_BIN, DATASKETCHER, PARSEFASTQ):
        assert os.access(exe, os.X_OK)
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    r = subprocess.run([DATASKETCHER, "-f", "/nonexistent.fastq", "-s", "8", "-k", "8", "-d", "/tmp/none.sig"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
    r = subprocess.run([TEST_BIN, "test_pminhasha_kmer_smallb"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "FAIL test_pminhasha_kmer_smallb" in r.stdout and "no CPU fallback" in r.stdout


@pytest.fixture(scope="module")
def mirror_results(host_programs):
    r = subprocess.run([TEST_BIN], capture_output=True, text=True, timeout=900)
    res = {}
    for line in r.stdout.splitlines():
        m = re.match(r"(ok|FAIL) (\w+)(: .*)?$", line)
        if m:
            res[m.group(2)] = (m.group(1) == "ok", line)
    return res, r


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_mirror_case(mirror_results, case):
    res, r = mirror_results
    assert case in res, "test program did not report %s (rc %s)\n%s\n%s" % (case, r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert res[case][0], res[case][1]


def _fastq(tmp_path, seed, n):
    rng = np.random.default_rng(seed)
    reads, lines = [], []
    for r in range(n):
        s = "".join(rng.choice(list("ACGT"), size=int(rng.integers(40, 600))))
        if r % 9 == 4:
            s = s[:10] + "N" + s[11:]
        else:
            reads.append(s.encode())
        lines += ["@r%d" % r, s, "+", "I" * len(s)]
    p = tmp_path / "reads.fastq"
    p.write_text("\n".join(lines) + "\n")
    return str(p), reads


@pytest.mark.gpu
def test_datasketcher_tool(host_programs, tmp_path, oracle):
    """datasketcher -f .. -k 8 -s 200 -d ..: the dump holds the oracle's ProbMinHash3a rows of the accepted reads"""
    fq, reads = _fastq(tmp_path, 11, 120)
    out = str(tmp_path / "out.sig")
    r = subprocess.run([DATASKETCHER, "-f", fq, "-k", "8", "-s", "200", "-d", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "number of non acgt sequences" in r.stderr
    rd = formats.SigSketchFileReader(out)
    assert (rd.get_kmer_size(), rd.get_signature_length(), rd.get_signature_size()) == (8, 200, 4)
    bases, off = oracle.concat(reads)
    p = A.SketchParams(A.ALGO_PROB3A, A.KMER32BIT, 8, 200, A.SIG_U32, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 0, 0, 0, 0, 0)
    assert np.array_equal(rd.read_all(), oracle.sketch(bases, off, p))
    # by blocks
    outb = str(tmp_path / "outb.sig")
    r = subprocess.run([DATASKETCHER, "-f", fq, "-k", "8", "-s", "20", "-d", outb, "-b", "100"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr
    rb = formats.SigBlockSketchFileReader(outb)
    assert (rb.sketch_size, rb.kmer_size, rb.block_size) == (20, 8, 100)
    pb = A.SketchParams(A.ALGO_PROB3A, A.KMER32BIT, 8, 20, A.SIG_U32, A.HASHER_NOHASH, A.FHASH_CANON_INVHASH, 100, 0, 0, 0, 0)
    bro = oracle.block_layout(off, 100)
    want = oracle.sketch(bases, off, pb)
    row = 0
    for i in range(len(reads)):
        numseq, blocks = rb.next()
        assert numseq == i and len(blocks) == int(bro[i + 1] - bro[i])
        for nb, sig in blocks:
            assert np.array_equal(sig, want[row])
            row += 1
    assert rb.next() is None and row == int(bro[-1])


@pytest.mark.gpu
def test_parsefastq_tool(host_programs, tmp_path, oracle):
    """parsefastq -f .. kmer --count -s 21: the dump holds the oracle's (canonical k-mer, count >= 2) records"""
    fq, reads = _fastq(tmp_path, 12, 60)
    reads2 = reads + reads[:20]  # make multiplicities
    with open(fq, "a") as f:
        for i, s in enumerate(reads[:20]):
            f.write("@again%d\n%s\n+\n%s\n" % (i, s.decode(), "I" * len(s)))
    r = subprocess.run([PARSEFASTQ, "-f", fq, "kmer", "--count", "-s", "21", "-t", "4", "--outdir", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    k, vals, counts = formats.load_kmer_counter(str(tmp_path / "reads.fastq.multi_kmer.bin"), 8)
    bases, off = oracle.concat(reads2)
    g = oracle.Counter(A.KMER64BIT, 21, 8, 1 << 20)
    g.add_reads(bases, off)
    gk, gc = g.dump(2)
    assert k == 21 and np.array_equal(vals, gk) and np.array_equal(counts, np.minimum(gc, 255))
    with open(str(tmp_path / "reads.fastq.multi_kmer.bin"), "rb") as f:
        assert struct.unpack("<I", f.read(4))[0] == formats.COUNTER_MULTIPLE


def _c_example():
    exe = os.path.join(ROOT, "examples", "sketch_c")
    subprocess.run(["make", "-s", "-C", ROOT, "examples/sketch_c"], check=True, capture_output=True, timeout=900)
    return exe


def test_c_example_builds_with_plain_gcc():
    """examples/sketch.c: the C-ABI from C99 (gcc, no C++ / HIP headers); without a device it stops with the library's error"""
    import torch
    exe = _c_example()
    assert os.access(exe, os.X_OK)
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_example_runs():
    r = subprocess.run([_c_example()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "= 1.000" in r.stdout


@pytest.mark.gpu
def test_parsefastq_tool_unique_branch(host_programs, tmp_path, oracle):
    """parsefastq -f .. kmer --unique: <file>.once_kmer.bin holds the oracle's once-16-mers with their positions"""
    fq, reads = _fastq(tmp_path, 13, 80)
    r = subprocess.run([PARSEFASTQ, "-f", fq, "kmer", "--unique", "--outdir", str(tmp_path)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr
    k, vals, numseq, numkmer = formats.load_once_kmers(str(tmp_path / "reads.fastq.once_kmer.bin"))
    bases, off = oracle.concat(reads)
    g = oracle.Counter(A.KMER16B32BIT, 16, 8, 1 << 20)
    g.add_reads(bases, off)
    wk, ws, wp = g.once_positions(bases, off)
    assert k == 16 and np.array_equal(vals, wk) and np.array_equal(numseq, ws) and np.array_equal(numkmer, wp)
    # and the Python mirror of the tool writes the same file
    from kmerutils_amd import parsefastq
    os.makedirs(str(tmp_path / "py"), exist_ok=True)
    assert parsefastq.main(["-f", fq, "--unique", "--outdir", str(tmp_path / "py")]) == 0
    assert (tmp_path / "py" / "reads.fastq.once_kmer.bin").read_bytes() == (tmp_path / "reads.fastq.once_kmer.bin").read_bytes()
