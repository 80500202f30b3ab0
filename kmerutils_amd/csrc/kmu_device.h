// kmu_device.h -- device-side primitives shared by every gfx950 kernel of libkmu.
//
// Everything here is integer / f64 arithmetic that must agree bit for bit with the reference (citations are
// file:line in jean-pierreBoth/kmerutils) or with the third-party crates it calls (probminhash, rand,
// rand_xoshiro: restated from their published algorithms, see DESIGN.md "unpinned").
// Compile with -ffp-contract=off: h = hbase + winv * x must not become an FMA.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kmu.h"

#define KMU_WAVE 64

namespace kmu {

// ---------------------------------------------------------------------------------------------------
// alphabet (src/base/alphabet.rs:119-127): A/a 0, C/c 1, G/g 2, T/t 3
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t code2b(uint32_t c) {
    uint32_t x = (c >> 1) & 3u; // A 0, C 1, T 2, G 3
    return x ^ (x >> 1);        // A 0, C 1, G 2, T 3
}
__device__ __forceinline__ bool is_acgt(uint32_t c) {
    uint32_t u = c & 0xDFu;
    return (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
}
// 16 ASCII bytes -> one packed word, first base in bits 31..30 (the MSB-first order of Sequence::new,
// src/base/sequence.rs:48-73, widened from a byte to a 32-bit word); `bad` gets one bit per invalid byte.
// Four bytes at a time (the byte-by-byte form -- code2b / is_acgt above -- cost 11 instructions per base, a tenth of the level-1
// kernel of the count and of the multiset kernels of the sketch; scripts/micro/swar_pack_check.c compares this form with the
// byte-wise rule over every byte value in every position):
// codes: x ^ (x >> 1) in every byte, then one multiplication moves the four 2-bit fields (bits 0, 8, 16, 24) to bits 30, 28, 26, 24.
__device__ __forceinline__ uint32_t codes4_ascii(uint32_t c) {
    uint32_t x = (c >> 1) & 0x03030303u;
    x ^= (x >> 1) & 0x01010101u;
    return (x * 0x40100401u) >> 24; // 1 + 2^10 + 2^20 + 2^30: no two partial products share a bit
}
// validity: with u = c & 0xDF, bits 1 and 2 of a byte select the one letter of ACGT that has them (A 00, C 01, G 11, T 10:
// 0x40 | the two bits | 0x10 and not 0x01 for T); the byte is valid iff it IS that letter.  A zero byte of the result = valid.
__device__ __forceinline__ uint32_t diff4_ascii(uint32_t c) {
    const uint32_t u = c & 0xDFDFDFDFu;
    const uint32_t t = (u >> 2) & ~(u >> 1) & 0x01010101u; // T
    const uint32_t e = 0x40404040u | (u & 0x06060606u) | (t << 4) | (t ^ 0x01010101u);
    return u ^ e;
}
// the non-zero bytes of d as four bits (byte 0 -> bit 0)
__device__ __forceinline__ uint32_t nonzero4(uint32_t d) {
    const uint32_t nz = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
    return (((nz >> 7) * 0x01020408u) >> 24) & 0xFu; // 2^3 + 2^10 + 2^17 + 2^24
}
__device__ __forceinline__ uint32_t pack16_ascii(uint4 v, uint32_t &bad) {
    const uint32_t out = (codes4_ascii(v.x) << 24) | (codes4_ascii(v.y) << 16) | (codes4_ascii(v.z) << 8) | codes4_ascii(v.w);
    const uint32_t d0 = diff4_ascii(v.x), d1 = diff4_ascii(v.y), d2 = diff4_ascii(v.z), d3 = diff4_ascii(v.w);
    bad = 0;
    if (d0 | d1 | d2 | d3) bad = nonzero4(d0) | (nonzero4(d1) << 4) | (nonzero4(d2) << 8) | (nonzero4(d3) << 12);
    return out;
}
// amino acids (src/aautils/kmeraa.rs:85-109): 5-bit codes, upper case only, Q = 15, no 14; 0 = invalid
__device__ __forceinline__ uint32_t code_aa(uint32_t c) {
    // index by (c - 'A'): A B C D E F G H I J K L M N O P Q R S T U V W X Y Z
    //                     1 0 2 3 4 5 6 7 8 0 9 10 11 12 0 13 15 16 17 18 0 19 20 0 21 0
    // as 5-bit fields of three constants (twelve letters each): a table in memory is a load per residue with a per-lane address --
    // twelve of them per k-mer in wave_step_kmers_aa, next to the twelve loads of the residues themselves
    const uint32_t i = c - 'A';
    const uint64_t t = i < 12u ? 0x52408398a418801ull : i < 24u ? 0x52609460f6818bull : 0x15ull;
    const uint32_t f = i < 12u ? i : i < 24u ? i - 12u : (i - 24u) & 1u; // (a shift below 64 whatever the byte)
    return i < 26u ? (uint32_t) (t >> (5u * f)) & 31u : 0u;
}

// ---------------------------------------------------------------------------------------------------
// k-mer values (A.2/A.3 of SURVEY.md)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t revcomp_val(uint64_t val, int k) {
    // kmer64bit.rs:83-96 / kmer32bit.rs:119-137 / kmer16b32bit.rs:43-54 on the right-aligned 2k-bit value
    uint64_t rc = ~val;
    rc = __brevll(rc);
    rc = ((rc & 0x5555555555555555ull) << 1) | ((rc & 0xAAAAAAAAAAAAAAAAull) >> 1);
    return rc >> (64 - 2 * k);
}

// A window of 48 bases b0 .. b47 (three code words; the first base in the top bits) and its reverse complement
// R = comp(b47) .. comp(b0), taken once per window (three 16-base words) instead of once per k-mer: the k-mer that starts at base j
// is bits of the window from base j on, its reverse complement (kmer64bit.rs:83-96) the 2 k bits of R that end 2 j bits above
// R's bottom.  (Level 1 of the count: a lane's window of a wave step; k_multiset_uq: a thread's run of positions.)
struct StepWin {
    uint64_t hi, rlo, kmask;
    uint32_t w2, rhi;
    int sh;
};
__device__ __forceinline__ uint32_t revcomp16(uint32_t w) {
    const uint32_t r = __brev(~w);
    return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}
__device__ __forceinline__ StepWin step_win(uint32_t w0, uint32_t w1, uint32_t w2, int k) {
    StepWin s;
    s.hi = ((uint64_t) w0 << 32) | w1;
    s.w2 = w2;
    s.rlo = ((uint64_t) revcomp16(w1) << 32) | revcomp16(w0);
    s.rhi = revcomp16(w2);
    s.sh = 64 - 2 * k;
    s.kmask = ~0ull >> (64 - 2 * k);
    return s;
}
// the k-mer at base j of the window (0 .. 15; a constant after unrolling, or a lane's own offset) and its reverse complement
__device__ __forceinline__ void step_val_rc(const StepWin &s, uint32_t j, uint64_t &val, uint64_t &rc) {
    val = ((s.hi << (2 * j)) | (((uint64_t) s.w2 << (2 * j)) >> 32)) >> s.sh;
    rc = (j ? (s.rlo >> (2 * j)) | ((uint64_t) s.rhi << (64 - 2 * j)) : s.rlo) & s.kmask;
}
__device__ __forceinline__ uint64_t step_canonical(const StepWin &s, uint32_t j) {
    uint64_t val, rc;
    step_val_rc(s, j, val, rc);
    return rc < val ? rc : val;
}

// probminhash::invhash (Thomas Wang hash32shift / hash64shift)
__device__ __forceinline__ uint32_t int32_hash(uint32_t key) {
    key = ~key + (key << 15);
    key = key ^ (key >> 12);
    key = key + (key << 2);
    key = key ^ (key >> 4);
    key = key * 2057u;
    key = key ^ (key >> 16);
    return key;
}
__device__ __forceinline__ uint64_t int64_hash(uint64_t key) {
    key = ~key + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

__device__ __forceinline__ uint64_t rotl64(uint64_t x, unsigned r) {
    r &= 63u;
    return (x << r) | (x >> ((64u - r) & 63u));
}

// ntHash seeds (src/base/nthash.rs:17-20), BASE_MAPPING_2B order [A,C,G,T]
__device__ __forceinline__ uint64_t nt_seed(uint32_t code) {
    const uint64_t s[4] = {0x3c8bfbb395c60474ull, 0x3193c18562a02b4cull, 0x20323ed082572324ull,
                           0x295549f54be24456ull};
    return s[code & 3u];
}
// forward / reverse seeds of the reference's *ASCII* table (nthash.rs:48-57): only 'A' and 'C' carry a seed,
// 'G' (71) and 'T' (84) hit zero entries; complement row: A -> SEED_T, C -> SEED_G, G/T -> 0.
__device__ __forceinline__ uint64_t nt8b_fwd(uint32_t code) {
    return code == 0 ? 0x3c8bfbb395c60474ull : (code == 1 ? 0x3193c18562a02b4cull : 0ull);
}
__device__ __forceinline__ uint64_t nt8b_rev(uint32_t code) {
    return code == 0 ? 0x295549f54be24456ull : (code == 1 ? 0x20323ed082572324ull : 0ull);
}
// canonical ntHash of a right-aligned k-mer value: kmer.rs:76-95 (2-bit seeds) or nthash.rs:214-228 (8-bit
// table, upper-case input).  Closed form per position; equals the rolled value (nthash.rs:333,379).
template <bool TABLE8B>
__device__ __forceinline__ uint64_t nthash_canonical(uint64_t val, int k) {
    uint64_t f = 0, r = 0;
    for (int i = 0; i < k; i++) {
        uint32_t b = (uint32_t) (val >> (2 * (k - 1 - i))) & 3u;
        if (TABLE8B) {
            f ^= rotl64(nt8b_fwd(b), (unsigned) (k - i - 1));
            r ^= rotl64(nt8b_rev(b), (unsigned) i);
        } else {
            f ^= rotl64(nt_seed(b), (unsigned) (k - i - 1));
            r ^= rotl64(nt_seed(3u - b), (unsigned) i);
        }
    }
    return f <= r ? f : r;
}

// ---------------------------------------------------------------------------------------------------
// k-mer type dependent pieces, selected at compile time.  `val` is always the right-aligned value.
// ---------------------------------------------------------------------------------------------------
struct KmerCfg {
    int kmer_type; // kmu_kmer_type
    int k;
    int fhash; // kmu_fhash
};

// raw `.0` from the value (KmerBuilder::build)
__device__ __forceinline__ uint64_t raw_from_val(int kmer_type, int k, uint64_t val) {
    return kmer_type == KMU_KMER32BIT ? (val | ((uint64_t) k << 28)) : val;
}
__device__ __forceinline__ bool is_u32_type(int kmer_type) {
    return kmer_type == KMU_KMER32BIT || kmer_type == KMU_KMER16B32BIT || kmer_type == KMU_KMERAA32BIT;
}

// fhash closure on a forward k-mer value (and its reverse complement, already computed by the caller for DNA)
__device__ __forceinline__ uint64_t apply_fhash(const KmerCfg &c, uint64_t val, uint64_t rcval) {
    const bool w32 = is_u32_type(c.kmer_type);
    switch (c.fhash) {
    case KMU_FHASH_IDENTITY_RAW: return raw_from_val(c.kmer_type, c.k, val);
    case KMU_FHASH_VALUE_MASKED: return val; // value is already masked to bits*k
    case KMU_FHASH_INVHASH_RAW: {
        uint64_t raw = raw_from_val(c.kmer_type, c.k, val);
        return w32 ? (uint64_t) int32_hash((uint32_t) raw) : int64_hash(raw);
    }
    default: break;
    }
    uint64_t canon = rcval < val ? rcval : val; // same k => Ord compares values (kmer32bit.rs:47-55)
    switch (c.fhash) {
    case KMU_FHASH_CANON_RAW: return raw_from_val(c.kmer_type, c.k, canon);
    case KMU_FHASH_CANON_VALUE: return canon;
    case KMU_FHASH_CANON_INVHASH: {
        uint64_t raw = raw_from_val(c.kmer_type, c.k, canon);
        return w32 ? (uint64_t) int32_hash((uint32_t) raw) : int64_hash(raw);
    }
    case KMU_FHASH_CANON_NTHASH: return nthash_canonical<false>(val, c.k);
    case KMU_FHASH_CANON_NTHASH_8B: return nthash_canonical<true>(val, c.k);
    default: return canon;
    }
}

// std::hash::Hasher::finish of the key: NoHashHasher (src/nohasher.rs:22-48), FnvHasher, or int64_hash
__device__ __forceinline__ uint64_t hasher_finish(int hasher, uint64_t v, bool w32) {
    if (hasher == KMU_HASHER_FNV1A) {
        uint64_t h = 0xcbf29ce484222325ull;
        int n = w32 ? 4 : 8;
        for (int i = 0; i < n; i++) {
            h ^= (v >> (8 * i)) & 0xFFull;
            h *= 0x100000001b3ull;
        }
        return h;
    }
    if (hasher == KMU_HASHER_INT64HASH) return int64_hash(v);
    if (w32) return (uint64_t) __builtin_bswap32((uint32_t) v);
    return __builtin_bswap64(v);
}

// ---------------------------------------------------------------------------------------------------
// RNG: Xoshiro256PlusPlus::seed_from_u64 (4 x SplitMix64) + rand's uniform samplers
// ---------------------------------------------------------------------------------------------------
struct Xoshiro {
    uint64_t s0, s1, s2, s3;
    __device__ __forceinline__ void seed(uint64_t x) {
        uint64_t z;
        x += 0x9e3779b97f4a7c15ull; z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; s0 = z ^ (z >> 31);
        x += 0x9e3779b97f4a7c15ull; z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; s1 = z ^ (z >> 31);
        x += 0x9e3779b97f4a7c15ull; z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; s2 = z ^ (z >> 31);
        x += 0x9e3779b97f4a7c15ull; z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; s3 = z ^ (z >> 31);
    }
    __device__ __forceinline__ uint64_t next() {
        uint64_t result = rotl64(s0 + s3, 23) + s0;
        uint64_t t = s1 << 17;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = rotl64(s3, 45);
        return result;
    }
    __device__ __forceinline__ uint32_t next_u32() { return (uint32_t) (next() >> 32); }
    // Uniform::<f64>::new(0.,1.): 52 mantissa bits in [1,2) minus 1
    __device__ __forceinline__ double unif01() {
        return __longlong_as_double((long long) ((next() >> 12) | 0x3FF0000000000000ull)) - 1.0;
    }
    __device__ __forceinline__ float unif01_f32() { return __uint_as_float((next_u32() >> 9) | 0x3F800000u) - 1.0f; }
    // Uniform::<usize>::new(low, high), high - 1 <= u32::MAX (always true here: ranges are sketch sizes)
    __device__ __forceinline__ uint32_t unif_index(uint32_t low, uint32_t high, bool rand08) {
        uint32_t range = high - low;
        if (rand08) {
            uint64_t r64 = range;
            uint64_t ints_to_reject = (0xFFFFFFFFFFFFFFFFull - r64 + 1ull) % r64;
            uint64_t zone = 0xFFFFFFFFFFFFFFFFull - ints_to_reject;
            for (;;) {
                uint64_t v = next();
                uint64_t hi = __umul64hi(v, r64), lo = v * r64;
                if (lo <= zone) return low + (uint32_t) hi;
            }
        }
        uint32_t thresh = (0u - range) % range;
        for (;;) {
            uint64_t m = (uint64_t) next_u32() * range;
            if ((uint32_t) m >= thresh) return low + (uint32_t) (m >> 32);
        }
    }
    // the same draw with the rejection bound of `range` taken from a table (index_bound(range, rand08)): it depends on the range alone
    __device__ static __forceinline__ uint64_t index_bound(uint32_t range, bool rand08) {
        if (rand08) return 0xFFFFFFFFFFFFFFFFull - (0xFFFFFFFFFFFFFFFFull - (uint64_t) range + 1ull) % (uint64_t) range; // zone
        return (uint64_t) ((0u - range) % range);                                                                       // thresh
    }
    __device__ __forceinline__ uint32_t unif_index_bound(uint32_t low, uint32_t high, uint64_t bound, bool rand08) {
        const uint32_t range = high - low;
        if (rand08) {
            for (;;) {
                const uint64_t v = next();
                const uint64_t hi = __umul64hi(v, (uint64_t) range), lo = v * (uint64_t) range;
                if (lo <= bound) return low + (uint32_t) hi;
            }
        }
        for (;;) {
            const uint64_t m = (uint64_t) next_u32() * range;
            if ((uint32_t) m >= (uint32_t) bound) return low + (uint32_t) (m >> 32);
        }
    }
};

// exp(x) - 1 on [0, ln 2]: the same fixed 22-term Horner as the oracle's expm1_small (no fused operations)
__device__ __forceinline__ double expm1_small(double x) {
    const double inv_fact[23] = {
        1.0, 1.0, 1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
        1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0, 1.0 / 87178291200.0,
        1.0 / 1307674368000.0, 1.0 / 20922789888000.0, 1.0 / 355687428096000.0, 1.0 / 6402373705728000.0,
        1.0 / 121645100408832000.0, 1.0 / 2432902008176640000.0, 1.0 / 51090942171709440000.0,
        1.0 / 1124000727777607680000.0};
    double acc = inv_fact[22];
    for (int i = 21; i >= 1; i--) acc = acc * x + inv_fact[i];
    return acc * x;
}

// probminhash ExpRestricted01: constants computed once on the host with libm (same expressions as the crate)
struct Exp01 {
    double lambda, c1, c2, c3;
};
__device__ __forceinline__ double exp01_sample(const Exp01 &e, Xoshiro &rng) {
    double x = e.c1 * rng.unif01();
    if (x < 1.0) return x;
    for (;;) {
        x = rng.unif01();
        if (x < e.c2) return x;
        double y = 0.5 * rng.unif01();
        if (y > 1.0 - x) {
            x = 1.0 - x;
            y = 1.0 - y;
        }
        if (x <= e.c3 * (1.0 - y)) return x;
        if (e.c1 * y <= 1.0 - x) return x;
        if (y * e.c1 * e.lambda <= expm1_small(e.lambda * (1.0 - x))) return x;
    }
}

// ---------------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int) (threadIdx.x & 63u); }
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global load and
// store of the wave (s_waitcnt vmcnt(0)), which serialises HBM streaming against the LDS phases of a kernel; use this
// one where the waves of a workgroup hand data to each other through LDS alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// inclusive prefix sum over the 64 lanes with DPP moves (row_shr 1/2/4/8 inside the rows of 16, then row_bcast15 /
// row_bcast31 across rows): six VALU instructions, no LDS crossbar round trips.  All lanes must be active.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t shfl_down_u32(uint32_t v, int d) { return (uint32_t) __shfl_down((int) v, d, 64); }
__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, int src) { return (uint32_t) __shfl((int) v, src, 64); }
// maximum of a 32-bit value over the 64 lanes, in every lane: DPP row shifts / broadcasts (the inclusive max scan of kmu_smer.hip)
// and one readlane -- no LDS round trips (wave_max_u64 below makes twelve: ds_bpermute)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false)); // row_shr:1
    v = mx(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false)); // row_shr:2
    v = mx(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false)); // row_shr:4
    v = mx(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false)); // row_shr:8
    v = mx(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false)); // row_bcast:15
    v = mx(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false)); // row_bcast:31
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint64_t o = ((uint64_t) (uint32_t) __shfl_xor((int) (v >> 32), d, 64) << 32) |
                     (uint32_t) __shfl_xor((int) (uint32_t) v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------
// Streaming one sequence through a wave: 64 lanes x 16 bases per step.
//
// Sequence data are read as aligned 16-byte chunks (global_load_dwordx4, 1 KiB per wave-instruction); every
// lane turns its chunk into one packed 32-bit word of 2-bit codes and picks up the two following words from
// its neighbours with wave shuffles, so a lane can roll the 16 k-mers that start inside its chunk (k <= 32)
// without LDS staging and without a barrier.
// ---------------------------------------------------------------------------------------------------
struct SeqView {
    const uint8_t *base; // ASCII: bases array; PACKED2: packed array
    uint64_t begin;      // ASCII: byte index of the first base; PACKED2: byte index of the first packed byte
    uint64_t len;        // number of bases
    uint64_t total;      // size in bytes of the whole `base` array (for guarded edge loads)
    int packed;          // 1 = KMU_INPUT_PACKED2
};

// one aligned 16-byte chunk at byte index `idx` (multiple of 16) with edge guards
__device__ __forceinline__ uint4 load_chunk16(const uint8_t *base, uint64_t idx, uint64_t total) {
    if (idx + 16 <= total) return *reinterpret_cast<const uint4 *>(base + idx);
    uint32_t w[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u}; // 'A' padding, masked by callers
    for (int i = 0; i < 16; i++)
        if (idx + i < total) {
            w[i >> 2] &= ~(0xFFu << (8 * (i & 3)));
            w[i >> 2] |= (uint32_t) base[idx + i] << (8 * (i & 3));
        }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Packed word number `widx` of the sequence's *aligned code stream*: ASCII: bases [16*widx, 16*widx+16) of the
// 16-byte aligned stream that starts at (begin & ~15); PACKED2: 4 packed bytes starting at (begin&~3)+4*widx.
// `bad` gets a mask of non-ACGT bytes that lie inside the sequence.
__device__ __forceinline__ uint32_t load_code_word(const SeqView &s, uint64_t widx, uint32_t &bad) {
    bad = 0;
    if (s.packed) {
        uint64_t b0 = (s.begin & ~3ull) + 4 * widx;
        uint32_t w = 0;
        if (b0 + 4 <= s.total) {
            w = __builtin_bswap32(*reinterpret_cast<const uint32_t *>(s.base + b0));
        } else {
            for (int i = 0; i < 4; i++)
                if (b0 + i < s.total) w |= (uint32_t) s.base[b0 + i] << (24 - 8 * i);
        }
        return w;
    }
    uint64_t a0 = (s.begin & ~15ull) + 16 * widx;
    if (a0 >= s.total) return 0;
    uint4 v = load_chunk16(s.base, a0, s.total);
    uint32_t b;
    uint32_t w = pack16_ascii(v, b);
    // keep only the bits of bytes inside [begin, begin+len)
    uint64_t lo = s.begin > a0 ? s.begin - a0 : 0;               // first valid byte in chunk
    uint64_t end = s.begin + s.len;
    uint64_t hi = end > a0 ? (end - a0 > 16 ? 16 : end - a0) : 0; // one past the last valid byte
    uint32_t m = (hi > lo) ? (uint32_t) (((1ull << hi) - 1ull) & ~((1ull << lo) - 1ull)) : 0u;
    bad = b & m;
    return w;
}
// ---- split form for prefetching (ASCII input): an aligned 16-byte chunk goes from HBM straight into LDS
// (global_load_lds_dwordx4: no VGPR is held while the request is in flight; lane l of the wave lands at lds_wave_base +
// 16 l), and is turned into a code word when the sequence's turn comes.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void chunk16_to_lds(const uint8_t *src, uint8_t *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
                                     (__attribute__((address_space(3))) void *) lds_wave_base, 16, 0, 0);
}
// true if word `widx` of the sequence's aligned stream is a whole chunk inside the array (chunk16_to_lds may fetch it)
__device__ __forceinline__ bool chunk_is_plain(const SeqView &s, uint64_t widx) {
    return !s.packed && (s.begin & ~15ull) + 16 * widx + 16 <= s.total;
}
__device__ __forceinline__ uint32_t code_word_from_chunk(const SeqView &s, uint64_t widx, u32x4 c, uint32_t &bad) {
    const uint64_t a0 = (s.begin & ~15ull) + 16 * widx;
    uint32_t b;
    const uint32_t w = pack16_ascii(make_uint4(c.x, c.y, c.z, c.w), b);
    const uint64_t lo = s.begin > a0 ? s.begin - a0 : 0;
    const uint64_t end = s.begin + s.len;
    const uint64_t hi = end > a0 ? (end - a0 > 16 ? 16 : end - a0) : 0;
    const uint32_t m = (hi > lo) ? (uint32_t) (((1ull << hi) - 1ull) & ~((1ull << lo) - 1ull)) : 0u;
    bad = b & m;
    return w;
}
// number of leading code slots of the aligned stream that precede the first base of the sequence
__device__ __forceinline__ uint32_t seq_lead(const SeqView &s) {
    return s.packed ? (uint32_t) (s.begin & 3ull) * 4u : (uint32_t) (s.begin & 15ull);
}

} // namespace kmu
