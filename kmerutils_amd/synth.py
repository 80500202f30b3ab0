"""Seeded synthetic read sets for the BASELINE configs.

Two generators with the same shape parameters:
  * numpy (host)  -- small parity sets for tests and for the CPU-baseline sample;
  * torch (device) -- full-size sets generated directly in HBM for bench.py (no dataset, no network).
Reads are upper-case ACGT only (the reference drops every read that holds another byte,
src/bin/datasketcher.rs:367-371, src/io.rs:41-48) and never empty.  Output container = concatenated ASCII bases +
uint64 offsets (n+1), the layout of include/kmu.h.
"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
AA20 = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
# UniProt-like residue frequencies (same order as AA20)
AA_FREQ = np.array([8.25, 1.37, 5.45, 6.75, 3.86, 7.07, 2.27, 5.96, 5.84, 9.66, 2.42, 4.06, 4.70, 3.93, 5.53, 6.56,
                    5.34, 6.87, 1.08, 2.92])

SEEDS = {"C1": 0xC1, "C2": 0xC2, "C3": 0xC3, "C4": 0xC4, "C5": 0xC5}


def _offsets(lens):
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens, dtype=np.uint64)
    return off


def uniform_reads(n_reads, read_len, seed):
    """C1: iid uniform ACGT reads of fixed length"""
    rng = np.random.default_rng(seed)
    bases = ACGT[rng.integers(0, 4, size=n_reads * read_len, dtype=np.uint8)]
    return bases, _offsets(np.full(n_reads, read_len, dtype=np.int64))


def ont_lengths(n_reads, rng, mean_target=None, mu=np.log(4600.0), sigma=0.75, lo=200, hi=200000):
    """log-normal ONT-shaped lengths clipped to [lo, hi]; optionally rescaled to a target mean"""
    L = np.exp(rng.normal(mu, sigma, size=n_reads))
    L = np.clip(L, lo, hi)
    if mean_target:
        L = np.clip(L * (mean_target / L.mean()), lo, hi)
    return np.maximum(L.astype(np.int64), 1)


def _revcomp_codes(codes):
    return (3 - codes)[::-1]


def genome_reads(n_reads, lens, genome_len, seed, sub=0.0, ins=0.0, dele=0.0):
    """reads sampled uniformly over both strands of an iid uniform genome with per-base errors (host, numpy)"""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    lens = np.asarray(lens, dtype=np.int64)
    out = np.empty(int(lens.sum()), dtype=np.uint8)
    pos = 0
    for L in lens:
        span = int(L * 1.1) + 8
        start = int(rng.integers(0, max(1, genome_len - span)))
        src = genome[start:start + span]
        if rng.integers(0, 2):
            src = _revcomp_codes(src)
        if ins > 0 or dele > 0:
            step = rng.choice(np.array([0, 1, 2]), size=L, p=[ins, 1.0 - ins - dele, dele])
            idx = np.minimum(np.cumsum(step), span - 1)
            read = src[idx]
            insmask = step == 0
            read[insmask] = rng.integers(0, 4, size=int(insmask.sum()), dtype=np.uint8)
        else:
            read = src[:L].copy()
        if sub > 0:
            m = rng.random(L) < sub
            read[m] = (read[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        out[pos:pos + L] = read
        pos += L
    return ACGT[out], _offsets(lens)


def illumina_reads(n_reads, genome_len, seed, read_len=150, sub=0.005):
    """C2 / C4 shape: fixed-length reads from a genome with substitution errors"""
    return genome_reads(n_reads, np.full(n_reads, read_len), genome_len, seed, sub=sub)


def ont_reads(n_reads, genome_len, seed, mean_len=None):
    """C3 shape: ONT-like lengths, 8 % errors (sub/ins/del 4/2/2)"""
    rng = np.random.default_rng(seed ^ 0x5EED)
    lens = ont_lengths(n_reads, rng, mean_target=mean_len)
    return genome_reads(n_reads, lens, genome_len, seed, sub=0.04, ins=0.02, dele=0.02)


def protein_seqs(n_seqs, seed, median=300, sigma=0.5, lo=20, hi=5000):
    """C5 shape: log-normal lengths, residues iid with UniProt-like frequencies"""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.exp(rng.normal(np.log(median), sigma, size=n_seqs)), lo, hi).astype(np.int64)
    p = AA_FREQ / AA_FREQ.sum()
    res = AA20[rng.choice(20, size=int(lens.sum()), p=p)]
    return res, _offsets(lens)


# ------------------------------------------------------------------------------------------------------------
# device-side generation (torch) for the full-size bench workloads
# ------------------------------------------------------------------------------------------------------------

def ont_reads_device(n_reads, total_bases, genome_len, seed, device, chunk_reads=4096, errors=(0.04, 0.02, 0.02),
                     fixed_len=None, read_seed=None):
    """Generate an ONT-shaped (or fixed-length) read set directly on `device`.

    Returns (bases uint8[total] cuda, offsets int64[n+1] cuda, lens int64 numpy).  Lengths come from the host
    generator (same distribution as ont_lengths, rescaled to total_bases / n_reads mean)."""
    import torch
    rng = np.random.default_rng(seed ^ 0x5EED)
    if fixed_len:
        lens = np.full(n_reads, int(fixed_len), dtype=np.int64)
    else:
        lens = ont_lengths(n_reads, rng, mean_target=total_bases / n_reads)
    offsets = np.zeros(n_reads + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(lens)
    total = int(offsets[-1])
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    genome = torch.randint(0, 4, (genome_len,), dtype=torch.uint8, device=device, generator=g)
    if read_seed is not None:  # same genome, different reads (one shard per rank)
        g.manual_seed(int(read_seed))
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    bases = torch.empty(total + 64, dtype=torch.uint8, device=device)
    sub, ins, dele = errors
    lens_t = torch.from_numpy(lens).to(device)
    off_t = torch.from_numpy(offsets).to(device)
    for r0 in range(0, n_reads, chunk_reads):
        r1 = min(n_reads, r0 + chunk_reads)
        b0, b1 = int(offsets[r0]), int(offsets[r1])
        nb = b1 - b0
        if nb == 0:
            continue
        cl = lens_t[r0:r1]
        rid = torch.repeat_interleave(torch.arange(r1 - r0, device=device), cl)
        within = torch.arange(nb, device=device) - (off_t[r0:r1] - b0)[rid]
        u = torch.rand(nb, device=device, generator=g)
        if ins > 0 or dele > 0:
            step = torch.ones(nb, dtype=torch.int64, device=device)
            step[u < ins] = 0
            step[u > 1.0 - dele] = 2
            cs = torch.cumsum(step, 0)
            first = (off_t[r0:r1] - b0)
            base_cs = (cs[first] - step[first])[rid]
            src_off = cs - base_cs - 1
            insmask = u < ins
        else:
            src_off = within
            insmask = None
        span = (cl.double() * 1.1).long() + 8
        starts = (torch.rand(r1 - r0, device=device, generator=g).double() *
                  (genome_len - span - 1).clamp(min=1).double()).long()
        strand = torch.rand(r1 - r0, device=device, generator=g) < 0.5
        src_off = torch.minimum(src_off, (span - 1)[rid])
        fwd_idx = starts[rid] + src_off
        rev_idx = starts[rid] + (span[rid] - 1 - src_off)
        st = strand[rid]
        code = torch.where(st, 3 - genome[rev_idx.clamp(0, genome_len - 1)], genome[fwd_idx.clamp(0, genome_len - 1)])
        if insmask is not None:
            rnd = torch.randint(0, 4, (nb,), dtype=torch.uint8, device=device, generator=g)
            code = torch.where(insmask, rnd, code)
        if sub > 0:
            u2 = torch.rand(nb, device=device, generator=g)
            add = torch.randint(1, 4, (nb,), dtype=torch.uint8, device=device, generator=g)
            code = torch.where(u2 < sub, (code + add) & 3, code)
        bases[b0:b1] = lut[code.long()]
    bases[total:] = ord("A")
    return bases[:total], off_t, lens
