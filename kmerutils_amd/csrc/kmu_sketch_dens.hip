// kmu_sketch_dens.hip -- one-permutation hashing with optimal / reverse-optimal densification for gfx950.
//
// Reference loop (src/sketching/setsketchert.rs:385-463, 521-599; AA: src/aautils/setsketchert.rs:482-746): for every
// k-mer occurrence `sminhash.sketch(&fhash(kmer))`, then `end_sketch()` ("it calls densification!"), then `get_hsketch()`.
// `sketch`: seed an RNG from hasher(value), draw r in [0,1) and a bin k in [0,m); bin k keeps its smallest r.
// `end_sketch`: bins no item fell into are filled from bins that were -- OptDens: an empty bin probes h(i, attempt) until
// it meets a filled one (Shrivastava 2017); RevOptDens: in rounds, every filled bin offers itself to bin h(j, round), an
// empty bin takes the smallest j (Mai et al. 2019).  (Inner arithmetic: crate probminhash::densminhash, not in the
// reference tree -- "parity unpinned", DESIGN.md section 5 says what is restated and what is this implementation's.)
//
// Device mapping.  The sketch is a per-bin minimum over items, so items are independent: one lane per k-mer occurrence,
// `ds_min_u64` on the order-preserving bit pattern of the (positive) value; no multiset, no staging.
//   k_oph_reads<false>  one 256-thread workgroup per sequence (queue), bins in LDS, densification in LDS, row out
//   k_oph_reads<true>   one signature for all sequences: every workgroup accumulates the sequences it takes, then merges
//                       its bins into one global row (atomicMin)
//   k_oph_long          a sequence too long for one workgroup (genomes): the whole grid walks its words, same merge
//   k_oph_finish        one workgroup: global row -> LDS -> densify -> signature row
// Densification in parallel, same result as the sequential statement: the "filled by an item" bitmap is fixed before it
// starts, so an OptDens bin's search does not depend on the order bins are visited in; a RevOptDens round resolves
// competing offers with an atomic minimum on the offering bin's index.
#include <algorithm>
#include <vector>

#include "kmu_ctx.hpp"
#include "kmu_stream.h"

namespace kmu {

struct DensArgs {
    const uint8_t *bases;
    const uint64_t *offsets;
    const uint64_t *packed_offsets;
    uint32_t n_seq;
    int packed;
    uint64_t total_bytes;
    KmerCfg cfg;
    int m;
    int hasher;
    int rand08;
    int f32;      // signature is f32 (bins hold f32 bit patterns)
    int val_w32;  // Kmer::Val is 32 bits
    int rev;      // reverse densification
    int hll;      // SetSketch registers instead of bins: maxima of k = floor(1 - log_b x), no densification
    int sig_bytes; // 2 / 4 / 8: width of a signature entry
    uint32_t q;   // SetSketch: registers are clamped to [0, q + 1]
    double inv_am, inv_ln_b; // SetSketch: 1 / (a m), 1 / ln b
    uint32_t idx_thresh;  // rand 0.9 Uniform<usize>(0, m): reject while lo < (2^32 - m) % m
    uint64_t idx_zone;    // rand 0.8 Uniform<usize>(0, m): accept while lo <= zone
    const void *hashed; // pre-hashed input (offsets count values), else null
    int hashed_bytes;
    uint32_t skip_longer; // sequences with more k-mers are left to k_oph_long
    uint32_t long_seq;    // k_oph_long: the sequence to walk
    uint64_t *row;        // global accumulation row (m bit patterns), k_oph_reads<true> / k_oph_long / k_oph_finish
    uint64_t out_row;     // k_oph_finish: signature row to write
    void *sig_out;
    uint32_t *queue;
    uint32_t *err;
};

__device__ __forceinline__ uint64_t oph_large_bits(int f32) {
    return f32 ? (uint64_t) __float_as_uint(4294967296.0f) : (uint64_t) __double_as_longlong(4294967295.0); // F::from(u32::MAX)
}

// h(bin, attempt) in [0, m): SplitMix64 finaliser of the pair, reduced by multiply-high (this implementation's choice)
__device__ __forceinline__ uint32_t dens_hash(uint32_t bin, uint32_t attempt, uint32_t m) {
    uint64_t z = (((uint64_t) bin << 32) | attempt) + 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (uint32_t) __umul64hi(z, (uint64_t) m);
}

__device__ __forceinline__ SeqView dens_view(const DensArgs &a, uint32_t r) {
    SeqView sv;
    sv.base = a.bases;
    sv.len = a.offsets[r + 1] - a.offsets[r];
    sv.packed = a.packed;
    if (a.packed) {
        sv.begin = a.packed_offsets[r];
        sv.total = a.total_bytes ? a.total_bytes
                                 : (a.packed_offsets[a.n_seq - 1] + (a.offsets[a.n_seq] - a.offsets[a.n_seq - 1] + 3) / 4);
    } else {
        sv.begin = a.offsets[r];
        sv.total = a.total_bytes ? a.total_bytes : a.offsets[a.n_seq];
    }
    return sv;
}

// Uniform<usize>(0, m) with the rejection bounds computed once on the host (the sketch size is fixed for the launch)
__device__ __forceinline__ uint32_t dens_draw_bin(const DensArgs &a, Xoshiro &rng) {
    if (a.rand08) {
        for (;;) {
            const uint64_t v = rng.next();
            const uint64_t hi = __umul64hi(v, (uint64_t) a.m), lo = v * (uint64_t) a.m;
            if (lo <= a.idx_zone) return (uint32_t) hi;
        }
    }
    for (;;) {
        const uint64_t mm = (uint64_t) rng.next_u32() * (uint32_t) a.m;
        if ((uint32_t) mm >= a.idx_thresh) return (uint32_t) (mm >> 32);
    }
}

// natural logarithm of a positive normal double from +, -, *, / only (the oracle carries the same few lines): x = 2^e f with
// f in (sqrt(1/2), sqrt(2)], log f = 2 s (1 + z/3 + ... + z^10/21), s = (f - 1) / (f + 1), z = s^2
__device__ __forceinline__ double kmu_log(double x) {
    uint64_t bits = (uint64_t) __double_as_longlong(x);
    int e = (int) ((bits >> 52) & 0x7FF) - 1023;
    double f = __longlong_as_double((long long) ((bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
    if (f > 1.4142135623730951) { f = f * 0.5; e += 1; }
    const double s = (f - 1.0) / (f + 1.0), z = s * s;
    double poly = 1.0 / 21.0;
    poly = poly * z + 1.0 / 19.0;
    poly = poly * z + 1.0 / 17.0;
    poly = poly * z + 1.0 / 15.0;
    poly = poly * z + 1.0 / 13.0;
    poly = poly * z + 1.0 / 11.0;
    poly = poly * z + 1.0 / 9.0;
    poly = poly * z + 1.0 / 7.0;
    poly = poly * z + 1.0 / 5.0;
    poly = poly * z + 1.0 / 3.0;
    poly = poly * z + 1.0;
    return (double) e * 0.6931471805599453 + 2.0 * s * poly;
}

// K_low of this workgroup's registers, refreshed by one wave (registers only grow, so the word only grows).
// Every lane of the wave calls this (uniform control flow).
__device__ __forceinline__ void hll_refresh_klow(const DensArgs &a, const uint64_t *hs, uint32_t *klow) {
    uint64_t mn = ~0ull;
    for (int i = lane_id(); i < a.m; i += 64) {
        const uint64_t v = __hip_atomic_load(&hs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (not volatile: that becomes a FLAT load)
        mn = v < mn ? v : mn;
    }
    mn = ~wave_max_u64(~mn);
    if (lane_id() == 0) atomicMax(klow, (uint32_t) mn);
}

// SetSketch (Ertl 2021, Algorithm 1, SetSketch1), one element per lane (`have`), the wave in lock step: ascending
// x_j = x_{j-1} + Exp(1) / (a m); k = clamp(floor(1 - log_b x_j)); a lane stops at its first k <= K_low (any lower bound of
// this workgroup's registers: later k are no larger); else a uniformly drawn register takes max(K_i, k).  While the
// registers are young every element runs long, so K_low is refreshed inside the loop (all lanes take part).
__device__ __forceinline__ void hll_wave_items(const DensArgs &a, uint64_t *hs, uint32_t *klow, bool have, uint64_t value) {
    Xoshiro rng;
    rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
    if (have) rng.seed(hasher_finish(a.hasher, value, a.val_w32 != 0));
    double x = 0.0;
    int j = 0;
    uint32_t round = 0;
    bool active = have;
    while (__any(active)) {
        if (active) {
            x += -kmu_log(1.0 - rng.unif01()) * a.inv_am;
            const double t = 1.0 - kmu_log(x) * a.inv_ln_b;
            uint32_t k = 0;
            if (t >= (double) a.q + 1.0) k = a.q + 1u;
            else if (t > 0.0) k = (uint32_t) t;
            if (k <= __hip_atomic_load(klow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) active = false;
            else {
                const uint32_t i = dens_draw_bin(a, rng);
                atomicMax((unsigned long long *) &hs[i], (unsigned long long) k);
                if (++j >= a.m) active = false;
            }
        }
        if ((++round & 31u) == 0u) hll_refresh_klow(a, hs, klow);
    }
}

// one item: r and bin from its own RNG stream, minimum into the bin
__device__ __forceinline__ void oph_item(const DensArgs &a, uint64_t *hs, uint64_t value) {
    Xoshiro rng;
    rng.seed(hasher_finish(a.hasher, value, a.val_w32 != 0));
    uint64_t bits;
    if (a.f32) bits = (uint64_t) __float_as_uint(rng.unif01_f32());
    else bits = (uint64_t) __double_as_longlong(rng.unif01());
    const uint32_t k = dens_draw_bin(a, rng);
    atomicMin((unsigned long long *) &hs[k], (unsigned long long) bits);
}

// the items of words [st0, st1) x 64 of one sequence, waves striding by `stride` steps from `first`
// (HLL: compile-time, so that the bins kernels do not carry the registers of the SetSketch loop)
template <bool HLL>
__device__ __forceinline__ uint32_t oph_walk(const DensArgs &a, const SeqView &sv, uint64_t *hs, uint32_t *klow, bool aa,
                                             uint64_t nk, uint64_t first, uint64_t stride) {
    uint32_t bad = 0;
    if constexpr (HLL) {
        // the values of a step are collected first (the visit is per lane and per position), then worked off slot by slot
        // with the whole wave in step -- hll_wave_items needs uniform control flow
        if (a.hashed_bytes) {
            for (uint64_t p0 = first * 64; p0 < nk; p0 += stride * 64) {
                const uint64_t p = p0 + lane_id();
                uint64_t v = 0;
                if (p < nk)
                    v = a.hashed_bytes == 4 ? (uint64_t) reinterpret_cast<const uint32_t *>(a.hashed)[sv.begin + p]
                                            : reinterpret_cast<const uint64_t *>(a.hashed)[sv.begin + p];
                hll_wave_items(a, hs, klow, p < nk, v);
                if (((p0 / (stride * 64)) & 15u) == 15u) hll_refresh_klow(a, hs, klow);
            }
            return 0;
        }
        const uint64_t st1 = aa ? (sv.len + 63) / 64 : (seq_num_words(sv) + 63) / 64;
        for (uint64_t st = first; st < st1; st += stride) {
            uint64_t vals[16];
            uint32_t mask = 0;
            if (aa) {
                bad |= wave_step_kmers_aa(sv, a.cfg.k, st, 0, nk, [&](uint64_t, uint64_t val, uint64_t rc) {
                    vals[0] = apply_fhash(a.cfg, val, rc);
                    mask = 1u;
                });
            } else {
                const uint64_t p0 = (st * 64 + (uint64_t) lane_id()) * 16; // position + lead of this lane's first base
                const uint32_t lead = seq_lead(sv);
                bad |= wave_step_kmers(sv, a.cfg.k, st, 0, nk, [&](uint64_t pos, uint64_t val, uint64_t rc) {
                    const uint32_t slot = (uint32_t) (pos + lead - p0);
                    vals[slot] = apply_fhash(a.cfg, val, rc);
                    mask |= 1u << slot;
                });
            }
#pragma unroll
            for (int slot = 0; slot < 16; slot++) {
                if (!__any(mask >> slot & 1u)) continue; // uniform
                hll_wave_items(a, hs, klow, (mask >> slot & 1u) != 0u, vals[slot]);
            }
            hll_refresh_klow(a, hs, klow);
        }
        return bad;
    }
    auto visit = [&](uint64_t, uint64_t val, uint64_t rc) { oph_item(a, hs, apply_fhash(a.cfg, val, rc)); };
    if (a.hashed_bytes) {
        for (uint64_t p = first * 64 + lane_id(); p < nk; p += stride * 64) {
            const uint64_t v = a.hashed_bytes == 4 ? (uint64_t) reinterpret_cast<const uint32_t *>(a.hashed)[sv.begin + p]
                                                   : reinterpret_cast<const uint64_t *>(a.hashed)[sv.begin + p];
            oph_item(a, hs, v);
        }
    } else if (aa) {
        for (uint64_t st = first; st < (sv.len + 63) / 64; st += stride) bad |= wave_step_kmers_aa(sv, a.cfg.k, st, 0, nk, visit);
    } else {
        const uint64_t st1 = (seq_num_words(sv) + 63) / 64; // all words: the tail that starts no k-mer is validated too
        for (uint64_t st = first; st < st1; st += stride) bad |= wave_step_kmers(sv, a.cfg.k, st, 0, nk, visit);
    }
    return bad;
}

// Densification of hs[0, m) in LDS by the whole workgroup.  filled: bitmap words; claim: m words (RevOptDens), all ones on
// entry and on exit; cnt: two words.  Every thread of the workgroup calls this.
__device__ __forceinline__ void oph_densify(const DensArgs &a, uint64_t *hs, uint32_t *filled, uint32_t *claim, uint32_t *cnt) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t m = (uint32_t) a.m;
    const uint64_t large = oph_large_bits(a.f32);
    const uint32_t nwords = (m + 31) / 32;
    if (tid == 0) cnt[0] = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t w = tid; w < nwords; w += nthreads) {
        uint32_t bits = 0;
        for (uint32_t b = 0; b < 32 && w * 32 + b < m; b++) bits |= (hs[w * 32 + b] != large ? 1u : 0u) << b;
        filled[w] = bits;
        mine += (uint32_t) __popc(bits);
    }
    if (mine) atomicAdd(&cnt[0], mine);
    __syncthreads();
    const uint32_t n_filled = cnt[0];
    if (n_filled == 0 || n_filled == m) return; // nothing to copy from / nothing to fill (uniform)
    auto is_filled = [&](uint32_t j) { return (filled[j >> 5] >> (j & 31)) & 1u; };
    if (!a.rev) {
        for (uint32_t i = tid; i < m; i += nthreads) {
            if (is_filled(i)) continue;
            for (uint32_t attempt = 1;; attempt++) {
                const uint32_t j = dens_hash(i, attempt, m);
                if (is_filled(j)) { hs[i] = hs[j]; break; } // hs[j] of a filled bin never changes here
            }
        }
        __syncthreads();
        return;
    }
    uint32_t left = m - n_filled; // uniform
    for (uint32_t round = 1; left > 0; round++) {
        if (tid == 0) cnt[1] = 0;
        // offers: the smallest offering bin wins an empty target (the sequential sweep visits j in increasing order)
        for (uint32_t j = tid; j < m; j += nthreads) {
            if (!is_filled(j)) continue;
            const uint32_t i = dens_hash(j, round, m);
            if (hs[i] == large) atomicMin(&claim[i], j);
        }
        __syncthreads();
        uint32_t got = 0;
        for (uint32_t j = tid; j < m; j += nthreads) {
            if (!is_filled(j)) continue;
            const uint32_t i = dens_hash(j, round, m);
            if (claim[i] == j) { hs[i] = hs[j]; got++; } // exactly one offering bin sees its own index
        }
        if (got) atomicAdd(&cnt[1], got);
        __syncthreads();
        for (uint32_t j = tid; j < m; j += nthreads) { // wipe the claims of this round
            if (!is_filled(j)) continue;
            claim[dens_hash(j, round, m)] = 0xFFFFFFFFu;
        }
        left -= cnt[1];
        __syncthreads();
    }
}

__device__ __forceinline__ void oph_store_row(const DensArgs &a, const uint64_t *hs, uint64_t row) {
    for (int t = threadIdx.x; t < a.m; t += blockDim.x) {
        if (a.sig_bytes == 2) reinterpret_cast<uint16_t *>(a.sig_out)[row * a.m + t] = (uint16_t) hs[t];
        else if (a.sig_bytes == 4) reinterpret_cast<uint32_t *>(a.sig_out)[row * a.m + t] = (uint32_t) hs[t];
        else reinterpret_cast<uint64_t *>(a.sig_out)[row * a.m + t] = hs[t];
    }
}

// neutral element of a bin / register, and the merge of a workgroup's array into the global row
__device__ __forceinline__ uint64_t oph_neutral(const DensArgs &a) { return a.hll ? 0ull : oph_large_bits(a.f32); }
__device__ __forceinline__ void oph_merge_to_row(const DensArgs &a, const uint64_t *hs) {
    const uint64_t neutral = oph_neutral(a);
    for (int s = threadIdx.x; s < a.m; s += blockDim.x) {
        if (hs[s] == neutral) continue;
        if (a.hll) atomicMax((unsigned long long *) &a.row[s], (unsigned long long) hs[s]);
        else atomicMin((unsigned long long *) &a.row[s], (unsigned long long) hs[s]);
    }
}

// LDS layout shared by the kernels: hs[m] | filled[(m + 31) / 32] | cnt[4] | claim[m] (RevOptDens only)
__device__ __forceinline__ void oph_lds(const DensArgs &a, uint8_t *smem, uint64_t *&hs, uint32_t *&filled, uint32_t *&cnt,
                                        uint32_t *&claim) {
    hs = reinterpret_cast<uint64_t *>(smem);
    filled = reinterpret_cast<uint32_t *>(hs + a.m);
    cnt = filled + (a.m + 31) / 32;
    claim = cnt + 4;
}

template <bool ALL, bool HLL>
__global__ void __launch_bounds__(256) k_oph_reads(DensArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint64_t *hs;
    uint32_t *filled, *cnt, *claim;
    oph_lds(a, smem, hs, filled, cnt, claim);
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int wave = tid >> 6, nwaves = nthreads >> 6;
    const bool aa = a.cfg.kmer_type == KMU_KMERAA32BIT || a.cfg.kmer_type == KMU_KMERAA64BIT;
    const uint64_t large = oph_neutral(a);
    for (int s = tid; s < a.m; s += nthreads) {
        hs[s] = large;
        if (a.rev && !ALL) claim[s] = 0xFFFFFFFFu;
    }
    if (tid == 0) { cnt[2] = atomicAdd(a.queue, 1u); cnt[3] = 0; } // cnt[3]: K_low of the registers (SetSketch)
    __syncthreads();
    for (;;) {
        const uint32_t r = cnt[2];
        __syncthreads(); // everyone holds r before thread 0 posts the next one
        if (r >= a.n_seq) break;
        uint32_t r_next = 0;
        if (tid == 0) r_next = atomicAdd(a.queue, 1u);
        const SeqView sv = dens_view(a, r);
        const uint64_t L = sv.len;
        const uint64_t nk = L >= (uint64_t) a.cfg.k ? L - a.cfg.k + 1 : 0;
        if (L == 0 && tid == 0 && !a.hashed_bytes) atomicOr(a.err, 8u);
        const bool skip = a.skip_longer && nk > (uint64_t) a.skip_longer; // k_oph_long's
        if (!skip) {
            uint32_t bad = 0;
            if (nk == 0 && !a.hashed_bytes) bad = wave_validate_seq(sv, wave, nwaves, aa);
            else bad = oph_walk<HLL>(a, sv, hs, &cnt[3], aa, nk, (uint64_t) wave, (uint64_t) nwaves);
            if (bad) atomicOr(a.err, aa ? DERR_BAD_AA : DERR_NON_ACGT);
        }
        __syncthreads();
        if (!ALL && !skip) {
            if (!a.hll) oph_densify(a, hs, filled, claim, cnt);
            oph_store_row(a, hs, (uint64_t) r);
            __syncthreads();
            for (int s = tid; s < a.m; s += nthreads) hs[s] = large;
            if (tid == 0) cnt[3] = 0;
        }
        if (tid == 0) cnt[2] = r_next;
        __syncthreads();
    }
    if (ALL) oph_merge_to_row(a, hs);
}

// one long sequence, walked by the whole grid; bins merged into a.row
template <bool HLL>
__global__ void __launch_bounds__(256) k_oph_long(DensArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint64_t *hs = reinterpret_cast<uint64_t *>(smem);
    uint32_t *klow = reinterpret_cast<uint32_t *>(hs + a.m);
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int wave = tid >> 6, nwaves = nthreads >> 6;
    const bool aa = a.cfg.kmer_type == KMU_KMERAA32BIT || a.cfg.kmer_type == KMU_KMERAA64BIT;
    const uint64_t large = oph_neutral(a);
    for (int s = tid; s < a.m; s += nthreads) hs[s] = large;
    if (tid == 0) *klow = 0;
    __syncthreads();
    const SeqView sv = dens_view(a, a.long_seq);
    const uint64_t nk = sv.len >= (uint64_t) a.cfg.k ? sv.len - a.cfg.k + 1 : 0;
    const uint32_t bad = oph_walk<HLL>(a, sv, hs, klow, aa, nk, (uint64_t) blockIdx.x * nwaves + wave, (uint64_t) gridDim.x * nwaves);
    if (bad) atomicOr(a.err, aa ? DERR_BAD_AA : DERR_NON_ACGT);
    __syncthreads();
    oph_merge_to_row(a, hs);
}

__global__ void __launch_bounds__(256) k_oph_fill(uint64_t *row, int m, uint64_t bits) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) row[t] = bits;
}

// global row -> densified signature row a.out_row
__global__ void __launch_bounds__(256) k_oph_finish(DensArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint64_t *hs;
    uint32_t *filled, *cnt, *claim;
    oph_lds(a, smem, hs, filled, cnt, claim);
    for (int s = threadIdx.x; s < a.m; s += blockDim.x) {
        hs[s] = a.row[s];
        if (a.rev) claim[s] = 0xFFFFFFFFu;
    }
    __syncthreads();
    if (!a.hll) oph_densify(a, hs, filled, claim, cnt);
    oph_store_row(a, hs, a.out_row);
}

__global__ void __launch_bounds__(1024) k_dens_max_len(const uint64_t *offsets, uint32_t n_seq, uint64_t *out) {
    uint64_t mx = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_seq; i += gridDim.x * blockDim.x) {
        const uint64_t L = offsets[i + 1] - offsets[i];
        mx = L > mx ? L : mx;
    }
    mx = wave_max_u64(mx);
    if (lane_id() == 0 && mx) atomicMax((unsigned long long *) out, (unsigned long long) mx);
}

static constexpr uint32_t DENS_LONG_KMERS = 1u << 20; // longer sequences are spread over the grid

// kmu_log on the host (for 1 / ln b): the same operations in the same order
static double host_log(double x) {
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int) ((bits >> 52) & 0x7FF) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double f;
    memcpy(&f, &bits, 8);
    if (f > 1.4142135623730951) { f = f * 0.5; e += 1; }
    const double s = (f - 1.0) / (f + 1.0), z = s * s;
    double poly = 1.0 / 21.0;
    poly = poly * z + 1.0 / 19.0;
    poly = poly * z + 1.0 / 17.0;
    poly = poly * z + 1.0 / 15.0;
    poly = poly * z + 1.0 / 13.0;
    poly = poly * z + 1.0 / 11.0;
    poly = poly * z + 1.0 / 9.0;
    poly = poly * z + 1.0 / 7.0;
    poly = poly * z + 1.0 / 5.0;
    poly = poly * z + 1.0 / 3.0;
    poly = poly * z + 1.0;
    return (double) e * 0.6931471805599453 + 2.0 * s * poly;
}

// OptDens / RevOptDens for every mode of kmu_sketch / kmu_sketch_hashed
int launch_dens(kmu_ctx *ctx, const kmu_sketch_params *p, const DevSeqs &ds, void *d_sig, uint32_t *d_err, const void *hashed,
                int hashed_bytes) {
    DensArgs a;
    memset(&a, 0, sizeof a);
    a.hashed = hashed;
    a.hashed_bytes = hashed_bytes;
    a.bases = ds.bases;
    a.offsets = ds.offsets;
    a.packed_offsets = ds.packed_offsets;
    a.n_seq = ds.n_seq;
    a.packed = ds.packed;
    a.total_bytes = ds.total_bytes;
    a.cfg = KmerCfg{p->kmer_type, hashed_bytes ? 1 : p->kmer_size, p->fhash};
    a.m = p->sketch_size;
    a.hasher = p->hasher;
    a.rand08 = (p->flags & KMU_FLAG_RAND08) ? 1 : 0;
    a.f32 = p->sig_type == KMU_SIG_F32;
    a.val_w32 = kmer_val_bytes(p->kmer_type) == 4;
    a.rev = p->algo == KMU_ALGO_REVOPTDENS;
    a.hll = p->algo == KMU_ALGO_HLL;
    a.sig_bytes = p->sig_type == KMU_SIG_U16 ? 2 : (p->sig_type == KMU_SIG_U32 || p->sig_type == KMU_SIG_F32) ? 4 : 8;
    a.idx_thresh = (0u - (uint32_t) a.m) % (uint32_t) a.m;
    a.idx_zone = 0xFFFFFFFFFFFFFFFFull - (0xFFFFFFFFFFFFFFFFull - (uint64_t) a.m + 1ull) % (uint64_t) a.m;
    if (a.hll) { // SetSketchParams of the context; 1 / ln b with the kernels' own logarithm (same formula on the host)
        a.q = ctx->hll.q;
        a.inv_am = 1.0 / (ctx->hll.a * (double) a.m);
        a.inv_ln_b = 1.0 / host_log(ctx->hll.b);
    }
    a.sig_out = d_sig;
    a.err = d_err;
    const bool all = p->mode == KMU_MODE_ALL_SEQS;
    const size_t lds_acc = (size_t) 8 * a.m + 16;                                            // bins only (k_oph_long)
    const size_t lds_full = ((size_t) 8 * a.m + 4 * ((size_t) (a.m + 31) / 32) + 16 + (a.rev ? (size_t) 4 * a.m : 0) + 15) & ~(size_t) 15;
    const size_t lds_max = 160 * 1024;
    if (lds_full > lds_max)
        return fail(ctx, KMU_E_UNSUPPORTED, "sketch_size %d: the bins of a densified sketch (%zu B) do not fit the LDS", a.m, lds_full);
    const void *fns[7] = {(const void *) k_oph_reads<false, false>, (const void *) k_oph_reads<true, false>,
                          (const void *) k_oph_reads<false, true>,  (const void *) k_oph_reads<true, true>,
                          (const void *) k_oph_long<false>,         (const void *) k_oph_long<true>,
                          (const void *) k_oph_finish};
    if (lds_full > 64 * 1024)
        for (const void *fn : fns)
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_max) != hipSuccess) {
                (void) hipGetLastError();
                return fail(ctx, KMU_E_UNSUPPORTED, "sketch_size %d needs %zu B of LDS", a.m, lds_full);
            }
    void *q, *row, *mx;
    KMU_TRY(dev_buf(ctx, "queue", 64, &q));
    KMU_HIP(ctx, hipMemsetAsync(q, 0, 64, ctx->stream));
    a.queue = (uint32_t *) q;
    KMU_TRY(dev_buf(ctx, "dens.row", (size_t) 8 * a.m + 64, &row));
    a.row = (uint64_t *) row;
    const uint64_t large = a.hll ? 0ull : a.f32 ? (uint64_t) 0x4F800000u /* 2^32 as f32 */ : 0x41EFFFFFFFE00000ull /* 4294967295.0 */;
    auto fill_row = [&]() {
        hipLaunchKernelGGL(k_oph_fill, dim3((a.m + 255) / 256), dim3(256), 0, ctx->stream, a.row, a.m, large);
    };
    // sequences too long for one workgroup: found on the host from the offsets (only if the longest one calls for it)
    std::vector<uint32_t> long_seqs;
    if (ds.n_seq) {
        KMU_TRY(dev_buf(ctx, "pmh.maxlen", 64, &mx));
        KMU_HIP(ctx, hipMemsetAsync(mx, 0, 8, ctx->stream));
        const uint32_t mgrid = (uint32_t) std::min<uint64_t>(((uint64_t) ds.n_seq + 1023) / 1024, (uint64_t) ctx->num_cus);
        hipLaunchKernelGGL(k_dens_max_len, dim3(mgrid ? mgrid : 1), dim3(1024), 0, ctx->stream, ds.offsets, ds.n_seq, (uint64_t *) mx);
        uint64_t max_len = 0;
        KMU_HIP(ctx, hipMemcpyAsync(&max_len, mx, 8, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (max_len > (uint64_t) DENS_LONG_KMERS + (uint64_t) a.cfg.k) {
            std::vector<uint64_t> h_off((size_t) ds.n_seq + 1);
            KMU_HIP(ctx, hipMemcpyAsync(h_off.data(), ds.offsets, ((size_t) ds.n_seq + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
            KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (uint32_t i = 0; i < ds.n_seq; i++) {
                const uint64_t L = h_off[i + 1] - h_off[i];
                if (L >= (uint64_t) a.cfg.k && L - a.cfg.k + 1 > DENS_LONG_KMERS) long_seqs.push_back(i);
            }
            a.skip_longer = DENS_LONG_KMERS;
        }
    }
    const int per_cu = (int) std::max<size_t>(1, std::min<size_t>(8, lds_max / lds_full));
    int grid = (int) std::min<uint64_t>(std::max<uint64_t>(ds.n_seq, 1), (uint64_t) ctx->num_cus * per_cu);
    if (all) fill_row();
    {
        KernelTimer t(ctx, "k_oph_reads");
        if (all && a.hll) hipLaunchKernelGGL((k_oph_reads<true, true>), dim3(grid), dim3(256), lds_full, ctx->stream, a);
        else if (all) hipLaunchKernelGGL((k_oph_reads<true, false>), dim3(grid), dim3(256), lds_full, ctx->stream, a);
        else if (a.hll) hipLaunchKernelGGL((k_oph_reads<false, true>), dim3(grid), dim3(256), lds_full, ctx->stream, a);
        else hipLaunchKernelGGL((k_oph_reads<false, false>), dim3(grid), dim3(256), lds_full, ctx->stream, a);
    }
    const int grid_long = ctx->num_cus * (int) std::max<size_t>(1, std::min<size_t>(8, lds_max / std::max<size_t>(lds_acc, 1024)));
    for (uint32_t i : long_seqs) {
        if (!all) fill_row();
        a.long_seq = i;
        KernelTimer t(ctx, "k_oph_long");
        if (a.hll) hipLaunchKernelGGL(k_oph_long<true>, dim3(grid_long), dim3(256), lds_acc, ctx->stream, a);
        else hipLaunchKernelGGL(k_oph_long<false>, dim3(grid_long), dim3(256), lds_acc, ctx->stream, a);
        if (!all) {
            a.out_row = i;
            hipLaunchKernelGGL(k_oph_finish, dim3(1), dim3(256), lds_full, ctx->stream, a);
        }
    }
    if (all && ctx->partial_out) { // the bins before densification are what other ranks' bins are merged with
        KMU_HIP(ctx, hipMemcpyAsync(ctx->partial_out, a.row, (size_t) 8 * a.m, hipMemcpyDeviceToDevice, ctx->stream));
    } else if (all) {
        a.out_row = 0;
        KernelTimer t(ctx, "k_oph_finish");
        hipLaunchKernelGGL(k_oph_finish, dim3(1), dim3(256), lds_full, ctx->stream, a);
    }
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

__global__ void __launch_bounds__(256) k_oph_merge(const uint64_t *parts, uint32_t n_parts, int m, int take_max, uint64_t *row) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    uint64_t best = parts[t];
    for (uint32_t i = 1; i < n_parts; i++) {
        const uint64_t v = parts[(uint64_t) i * m + t];
        best = (take_max ? v > best : v < best) ? v : best;
    }
    row[t] = best;
}

// bins of several ranks (device memory, n_parts x m patterns) -> one densified signature row
int launch_dens_merge(kmu_ctx *ctx, const kmu_sketch_params *p, const uint64_t *parts, uint32_t n_parts, void *d_sig) {
    DensArgs a;
    memset(&a, 0, sizeof a);
    a.m = p->sketch_size;
    a.f32 = p->sig_type == KMU_SIG_F32;
    a.rev = p->algo == KMU_ALGO_REVOPTDENS;
    a.hll = p->algo == KMU_ALGO_HLL;
    a.sig_bytes = p->sig_type == KMU_SIG_U16 ? 2 : (p->sig_type == KMU_SIG_U32 || p->sig_type == KMU_SIG_F32) ? 4 : 8;
    a.sig_out = d_sig;
    const size_t lds_full = ((size_t) 8 * a.m + 4 * ((size_t) (a.m + 31) / 32) + 16 + (a.rev ? (size_t) 4 * a.m : 0) + 15) & ~(size_t) 15;
    if (lds_full > 160 * 1024) return fail(ctx, KMU_E_UNSUPPORTED, "sketch_size %d: the bins do not fit the LDS", a.m);
    if (lds_full > 64 * 1024 &&
        hipFuncSetAttribute((const void *) k_oph_finish, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        (void) hipGetLastError();
        return fail(ctx, KMU_E_UNSUPPORTED, "sketch_size %d needs %zu B of LDS", a.m, lds_full);
    }
    void *row;
    KMU_TRY(dev_buf(ctx, "dens.row", (size_t) 8 * a.m + 64, &row));
    a.row = (uint64_t *) row;
    hipLaunchKernelGGL(k_oph_merge, dim3((a.m + 255) / 256), dim3(256), 0, ctx->stream, parts, n_parts, a.m, a.hll, a.row);
    a.out_row = 0;
    hipLaunchKernelGGL(k_oph_finish, dim3(1), dim3(256), lds_full, ctx->stream, a);
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

} // namespace kmu
