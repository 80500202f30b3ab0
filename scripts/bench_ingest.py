#!/usr/bin/env python3
"""diagnostics: throughput of kmu_ingest_fastq on a synthetic FASTQ text built on the device from the bench read set"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kmerutils_amd import lib, synth

dev = torch.device("cuda:0")
n_reads, total = int(os.environ.get("READS", 200_000)), float(os.environ.get("BASES", 1.2e9))
bases, offsets, lens = synth.ont_reads_device(n_reads, total, 50_000_000, 0xC3, dev)
bases = bases[:int(offsets[-1].item())]
L = (offsets[1:] - offsets[:-1]).to(torch.int64)
rec = 2 * L + 7                      # "@r\n" seq "\n+\n" qual "\n"
start = torch.cumsum(rec, 0) - rec
text = torch.full((int(rec.sum().item()),), ord("I"), dtype=torch.uint8, device=dev)
def put(pos, ch): text[pos] = ch
put(start, ord("@")); put(start + 1, ord("r")); put(start + 2, 10)
put(start + 3 + L, 10); put(start + 4 + L, ord("+")); put(start + 5 + L, 10); put(start + 6 + 2 * L, 10)
# sequence bytes: for every base its destination = its index + 3 * (read + 1) + sum of earlier read lengths + ...
rid = torch.repeat_interleave(torch.arange(len(L), device=dev), L)
idx = torch.arange(bases.shape[0], device=dev)
dst = idx - offsets[:-1].to(torch.int64)[rid] + start[rid] + 3
text[dst] = bases
# every 50th read gets an N
bad = start[::50] + 3 + (L[::50] // 2)
text[bad] = ord("N")
torch.cuda.synchronize()
ctx = lib.Context(0)
ctx.profile_enable(True)
for it in range(3):
    t0 = time.perf_counter()
    b, o, info = ctx.ingest_fastq(text)
    ctx.synchronize()
    dt = time.perf_counter() - t0
print(json.dumps({"text_GB": text.shape[0] / 1e9, "records": int(info.n_records), "kept": int(info.n_kept),
                  "bad_reads": int(info.nb_bad_reads), "seconds": dt, "text_GBps": text.shape[0] / dt / 1e9,
                  "kernels_ms": {k: round(v[1] / v[0], 3) for k, v in ctx.profile_get().items()}}))

# ---- FASTA: the same bases as a genome-style file, contigs of ~1 Mbase wrapped at 60 columns -------------------------------
W = 60
nb = bases.shape[0] - bases.shape[0] % W
seq = bases[:nb].clone()
contig = int(os.environ.get("CONTIG", 1_000_020)) // W * W        # whole lines per contig
n_contigs = (nb + contig - 1) // contig
lines = seq.view(-1, W)
body = torch.cat([lines, torch.full((lines.shape[0], 1), 10, dtype=torch.uint8, device=dev)], 1).reshape(-1)  # 61 bytes per line
hdr = torch.tensor(list(b">contig\n"), dtype=torch.uint8, device=dev)
per = contig // W * (W + 1)
parts = []
for c in range(n_contigs):
    parts += [hdr, body[c * per:(c + 1) * per]]
fa = torch.cat(parts)
del body, parts
torch.cuda.synchronize()
ctx.profile_reset()
for it in range(3):
    t0 = time.perf_counter()
    b, o, info = ctx.ingest_fastx(fa)
    ctx.synchronize()
    dt = time.perf_counter() - t0
assert int(info.n_records) == n_contigs and int(info.kept_bases) + int(info.n_bases - info.kept_bases) == nb
print(json.dumps({"fasta_GB": fa.shape[0] / 1e9, "records": int(info.n_records), "kept": int(info.n_kept),
                  "lines": int(lines.shape[0] + n_contigs), "seconds": dt, "text_GBps": fa.shape[0] / dt / 1e9,
                  "kernels_ms": {k: round(v[1] / v[0], 3) for k, v in ctx.profile_get().items()}}))
