// kmu_smer.h -- the minimizer owner of a k-mer and the super-k-mer record: what a distributed counter puts on the wire.
//
// Reference shape (src/base/kmercount.rs:881-974): one producer sends every canonical k-mer to the thread that owns it
// (`int64_hash(kmer) % n`, :412-420, :942), one message per k-mer.  Between GPUs a message per k-mer is 8 bytes per
// occurrence -- 30.5 GB per rank and step on the headline workload.  Here the owner of a k-mer is a function of its
// MINIMIZER (the canonical m-mer of smallest hash among the w = k - m + 1 it contains: both strands of a k-mer hold the
// same canonical m-mers, so they agree), consecutive k-mers of a read mostly share it, and a run of consecutive k-mers
// with the same owner travels as its BASES: one 12-byte record = up to 16 k-mers (46 bases at 2 bits + 4 bits of
// length), ~1.35 bytes per k-mer at k = 31.  The receiver expands a record into canonical k-mers with the window
// arithmetic of every other kernel of the path.  `int64_hash(kmer) % n` stays selectable (KMU_COUNT_OWNER_HASH).
//
// Everything that decides an owner is here, once, for the device kernels, the host entry point kmu_kmer_owner_minimizer
// and (restated base by base) the checker under tests/.
#pragma once

#include <stdint.h>

#include "../../include/kmu.h"

#ifndef KMU_HD
#ifdef __HIPCC__
#define KMU_HD __host__ __device__ __forceinline__
#else
#define KMU_HD inline
#endif
#endif

namespace kmu {

// Window shapes: w is one of three values so that the sliding minimum of the grouping kernels has three compile-time forms;
// m = k - w + 1 stays in 9 .. 15 (an m-mer is at most 30 bits).
struct SmerCfg {
    int k, m, w;
};
KMU_HD SmerCfg smer_cfg(int k) {
    SmerCfg c;
    c.k = k;
    c.w = k >= 29 ? 21 : (k >= 24 ? 16 : 9);
    c.m = k - c.w + 1;
    return c;
}
// minimizer owners exist for Kmer64bit with 17 <= k <= 31 (Kmer32bit / Kmer16b32bit: k <= 16 leaves no room for a window)
KMU_HD bool smer_supported(int kmer_type, int k) { return kmer_type == KMU_KMER64BIT && k >= 17 && k <= 31; }

// order of the canonical m-mers: a bijective mix of the 32-bit value (equal hashes = equal m-mers, so the minimum names one
// m-mer); the xor keeps poly-A (value 0) from being everybody's minimizer
KMU_HD uint32_t smer_mix(uint32_t x) {
    x ^= 0x5BD1E995u;
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    return x;
}
KMU_HD uint32_t smer_brev32(uint32_t x) {
#ifdef __HIP_DEVICE_COMPILE__
    return __brev(x);
#else
    x = (x >> 16) | (x << 16);
    x = ((x & 0xFF00FF00u) >> 8) | ((x & 0x00FF00FFu) << 8);
    x = ((x & 0xF0F0F0F0u) >> 4) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x & 0xCCCCCCCCu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
    return x;
#endif
}
// hash of the canonical form of the m-mer `f` (right-aligned, 2 m bits; A0 C1 G2 T3, first base in the high bits)
KMU_HD uint32_t smer_hash(uint32_t f, int m) {
    uint32_t r = smer_brev32(~f); // complement, base order reversed, the two bits of a base swapped ...
    r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u); // ... and swapped back
    r >>= 32 - 2 * m;
    return smer_mix(r < f ? r : f);
}
// owner of a minimizer hash in an n-way pool, and the duplication sample (a sample by KEY: every occurrence of a k-mer
// has the same minimizer), taken from other bits of the hash
KMU_HD uint32_t smer_owner_of(uint32_t minhash, uint32_t n_parts) { return ((minhash * 0xC2B2AE35u) >> 16) % n_parts; }
KMU_HD bool smer_sampled(uint32_t minhash, uint32_t shift) { return shift == 0u || ((minhash * 0x27D4EB2Fu) >> (32u - shift)) == 0u; }

// minimizer hash of a k-mer value (right-aligned 2 k bits; forward or canonical: the same set of canonical m-mers)
KMU_HD uint32_t smer_minhash_of_kmer(uint64_t v, int k) {
    const SmerCfg c = smer_cfg(k);
    const uint32_t mask = (1u << (2 * c.m)) - 1u;
    uint32_t mn = 0xFFFFFFFFu;
    for (int i = 0; i < c.w; i++) {
        const uint32_t h = smer_hash((uint32_t) (v >> (2 * (k - c.m - i))) & mask, c.m);
        mn = h < mn ? h : mn;
    }
    return mn;
}
KMU_HD uint32_t smer_owner_of_kmer(uint64_t v, int k, uint32_t n_parts) { return smer_owner_of(smer_minhash_of_kmer(v, k), n_parts); }

// ---- the record: 96 bits = three 32-bit words; the first base in bits 31..30 of word 0 (the code-word order of every kernel),
// n_kmers + k - 1 <= 46 bases, then zeros; bits 3..0 of word 2 = n_kmers - 1.
static constexpr uint32_t SMER_REC_BYTES = 12;
static constexpr uint32_t SMER_REC_KMERS = 16;
struct SmerRec {
    uint32_t w[3];
};
KMU_HD uint32_t smer_rec_kmers(const SmerRec &r) { return (r.w[2] & 15u) + 1u; }
// k-mer j (< smer_rec_kmers) of a record, forward value
KMU_HD uint64_t smer_rec_kmer(const SmerRec &r, int k, uint32_t j) {
    const uint64_t hi = ((uint64_t) r.w[0] << 32) | r.w[1];
    const uint64_t v = j ? (hi << (2 * j)) | (((uint64_t) r.w[2] << (2 * j)) >> 32) : hi;
    return v >> (64 - 2 * k);
}

} // namespace kmu
