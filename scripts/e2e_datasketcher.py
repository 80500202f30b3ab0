#!/usr/bin/env python3
"""End to end like the README's timing (`datasketcher -f x.fastq -k 8 -s 200 -d out`, 746 333 ONT reads / 4.4 Gbases:
51 s on the reference's 8-thread laptop): a FASTQ file of that shape is written to /tmp (not timed), then the C++ tool
(kmerutils_amd/bin/datasketcher) runs on it as a child process and is timed from outside.
READS / BASES scale the file; the dump is checked against an in-process device sketch of the same reads."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from kmerutils_amd import _abi as A
from kmerutils_amd import formats, lib, synth

dev = torch.device("cuda:0")
n_reads, total = int(os.environ.get("READS", 746_333)), float(os.environ.get("BASES", 4.38e9))
k, m = int(os.environ.get("K", 8)), int(os.environ.get("M", 200))
bases, offsets, lens = synth.ont_reads_device(n_reads, total, 100_000_000, 0xC3, dev)
nb = int(offsets[-1].item())
bases = bases[:nb]
L = (offsets[1:] - offsets[:-1]).to(torch.int64)
rec = 2 * L + 7                                   # "@r\n" seq "\n+\n" qual "\n"
start = torch.cumsum(rec, 0) - rec
text = torch.full((int(rec.sum().item()),), ord("I"), dtype=torch.uint8, device=dev)
text[start] = ord("@"); text[start + 1] = ord("r"); text[start + 2] = 10
text[start + 3 + L] = 10; text[start + 4 + L] = ord("+"); text[start + 5 + L] = 10; text[start + 6 + 2 * L] = 10
step = 1 << 28                                    # the sequence bytes, in slabs (index arrays are 8 bytes per base)
off64 = offsets.to(torch.int64)
for b0 in range(0, nb, step):
    b1 = min(nb, b0 + step)
    idx = torch.arange(b0, b1, device=dev)
    rid = torch.searchsorted(off64, idx, right=True) - 1
    text[idx - off64[rid] + start[rid] + 3] = bases[b0:b1]
    del idx, rid
fq = "/tmp/kmu_e2e.fastq"
t0 = time.perf_counter()
text.cpu().numpy().tofile(fq)
t_write = time.perf_counter() - t0
text_bytes = int(text.shape[0])
del text
torch.cuda.synchronize()

exe = os.path.join(ROOT, "kmerutils_amd", "bin", "datasketcher")
out = "/tmp/kmu_e2e.sig"
runs = []
for it in range(int(os.environ.get("RUNS", 4))):
    t0 = time.perf_counter()
    r = subprocess.run([exe, "-f", fq, "-k", str(k), "-s", str(m), "-d", out], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr
    runs.append({"wall_s": dt, "tool": [x.strip() for x in r.stderr.strip().splitlines() if "elapsed" in x]})

# the dump against an in-process sketch of the same reads
ctx = lib.Context(0)
p = A.SketchParams(A.ALGO_PROB3A, A.KMER32BIT, k, m, A.SIG_U32, 0, A.FHASH_CANON_INVHASH, 0, 0, A.INPUT_ASCII, A.MEM_DEVICE, 0)
sig = ctx.sketch(bases, offsets, p)
ctx.synchronize()
rd = formats.SigSketchFileReader(out)
rows = rd.read_all()
same = bool(rows.shape == tuple(sig.shape) and np.array_equal(rows, sig.cpu().numpy().view(np.uint32)))
os.remove(fq)
os.remove(out)
print(json.dumps({"reads": n_reads, "bases": nb, "fastq_GB": text_bytes / 1e9, "write_fastq_s": t_write, "k": k, "m": m,
                  "runs": runs, "Gbases_per_s_best": nb / min(x["wall_s"] for x in runs) / 1e9,
                  "dump_equals_device_sketch": same, "reference_README_s": 51.0}))
