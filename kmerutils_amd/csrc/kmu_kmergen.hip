// kmu_kmergen.hip -- the rest of the KmerGenerationPattern surface and of nthash.rs on gfx950.
//
//  * kmu_kmer_distribution: KmerGenerationPattern::generate_kmer_distribution (src/base/kmergenerator.rs:130; impls :245-269,
//    :339-366, :447-489; amino acids src/aautils/kmeraa.rs:752-778,845-868): the distinct k-mers of every sequence with their
//    multiplicities.  The reference fills an FnvHashMap while it iterates; here the fhash values of all k-mers are written
//    compact (the kernels of the all-sequences sketch), and work items (sequence, hash pass) build one small exact hash table in
//    LDS each (ds_cmpst_b64 + ds_add): a pass holds the keys whose hash falls into its slice, so a key is counted by exactly
//    one item and the counts are exact for every sequence length.
//  * kmu_nthash: nthash.rs:153-287 and the NtHash trait of 2-bit k-mers (src/base/kmer.rs:45-145) per k-mer position: forward,
//    reverse-complement and canonical hash, strand, and the multi-hash expansion from_one_hash_val_to_mult_hash (:63-72).
//    O(1) per position: every lane takes the XOR-prefix of the rotated seeds of its 16 bases, a wave-wide XOR scan (DPP) gives
//    the prefix at every lane start, the hash of a lane's first k-mer is the difference of two prefixes rotated back, and its
//    other 15 positions roll (nthash_cycle_8b / nthash_rcomp_cycle_8b, nthash.rs:172-176,198-202).
#include <algorithm>
#include <vector>

#include "kmu_ctx.hpp"
#include "kmu_smer.hpp"
#include "kmu_stream.h"

namespace kmu {

__global__ void k_nk_scan(const uint64_t *offsets, uint32_t n_seq, int k, uint64_t *koff, uint32_t *err);
__global__ void k_seq_hashes_compact(const uint8_t *bases, const uint64_t *offsets, const uint64_t *packed_offsets, uint32_t n_seq,
                                     int packed, uint64_t total, KmerCfg cfg, const uint64_t *koff, uint64_t *out, uint32_t *err,
                                     int spread);

// ------------------------------------------------------------------------------------------------------------------------
// k-mer distribution
// ------------------------------------------------------------------------------------------------------------------------
static constexpr uint32_t DIST_SLOTS = 4096;     // 48 KiB of LDS per workgroup: three workgroups per CU
static constexpr uint32_t DIST_PASS_KEYS = 1024; // k-mer occurrences aimed at per pass (load <= 1/4 when all are distinct)
static constexpr int DIST_THREADS = 512;
static constexpr uint64_t DKEY_EMPTY = 0xFFFFFFFFFFFFFFFFull;

__device__ __forceinline__ uint32_t dist_mix(uint64_t key) {
    uint32_t x = (uint32_t) key ^ (uint32_t) (key >> 32);
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    return x;
}

// passes[i] = number of hash passes of sequence i
__global__ void __launch_bounds__(256) k_dist_plan(const uint64_t *koff, uint32_t n_seq, uint32_t *passes) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_seq; i += gridDim.x * blockDim.x) {
        const uint64_t nk = koff[i + 1] - koff[i];
        passes[i] = nk ? (uint32_t) std::min<uint64_t>((nk + DIST_PASS_KEYS - 1) / DIST_PASS_KEYS, 1u << 24) : 0u;
    }
}

// One work item = (sequence, pass): the keys of the sequence with pass_of(key) == pass are counted in an LDS table and
// appended to the sequence's range of the temporary lists (tmp_k / tmp_c [koff[i] ..)), claimed with one atomic on nd[i].
// An item whose keys do not fit (very skewed hash) is redone in 2, 4, 8 ... sub-passes on further hash bits.
__global__ void __launch_bounds__(DIST_THREADS) k_kmer_dist(const uint64_t *vals, const uint64_t *koff, const uint64_t *item_off,
                                                            uint32_t n_seq, uint64_t n_items, uint64_t *tmp_k, uint32_t *tmp_c,
                                                            uint32_t *nd, unsigned long long *queue, uint32_t *err) {
    __shared__ uint64_t lk[DIST_SLOTS];
    __shared__ uint32_t lc[DIST_SLOTS];
    __shared__ uint32_t wtot[DIST_THREADS / 64];
    __shared__ uint64_t sh_item;
    __shared__ uint32_t sh_seq, sh_flag, sh_ones, sh_base;
    const uint32_t tid = threadIdx.x, lane = (uint32_t) lane_id(), wave = tid >> 6;
    for (;;) {
        if (tid == 0) {
            const uint64_t it = atomicAdd(queue, 1ull);
            sh_item = it;
            if (it < n_items) { // largest i with item_off[i] <= it
                uint32_t lo = 0, hi = n_seq;
                while (hi - lo > 1) {
                    const uint32_t mid = lo + (hi - lo) / 2;
                    if (item_off[mid] <= it) lo = mid; else hi = mid;
                }
                sh_seq = lo;
            }
        }
        __syncthreads();
        const uint64_t item = sh_item;
        if (item >= n_items) break;
        const uint32_t seq = sh_seq;
        const uint64_t b0 = koff[seq], nk = koff[seq + 1] - b0;
        const uint32_t P = (uint32_t) (item_off[seq + 1] - item_off[seq]), pass = (uint32_t) (item - item_off[seq]);
        // Sub-passes of this item: (S, sub) holds the keys with ((h2 >> 20) & (S - 1)) == sub.  The item starts as (1, 0); a
        // sub-pass whose keys do not fit the table is split into (2S, sub) and (2S, sub + S) -- nothing of it was appended.
        uint32_t stk[32]; // uniform; S in the high half, sub in the low half
        int sp = 0;
        stk[sp++] = (1u << 16) | 0u;
        while (sp > 0) {
            const uint32_t top = stk[--sp];
            const uint32_t S = top >> 16, sub = top & 0xFFFFu;
            for (uint32_t s = tid; s < DIST_SLOTS; s += DIST_THREADS) { lk[s] = DKEY_EMPTY; lc[s] = 0; }
            if (tid == 0) { sh_flag = 0; sh_ones = 0; }
            __syncthreads();
            for (uint64_t j = tid; j < nk; j += DIST_THREADS) {
                const uint64_t key = vals[b0 + j];
                const uint32_t h = dist_mix(key);
                if (P > 1 && (uint32_t) (((uint64_t) h * P) >> 32) != pass) continue;
                const uint32_t h2 = h * 0xC2B2AE35u;
                if (((h2 >> 20) & (S - 1)) != sub) continue;
                if (key == DKEY_EMPTY) { atomicAdd(&sh_ones, 1u); continue; } // the one value the empty marker shadows
                uint32_t off = h2 & (DIST_SLOTS - 1);
                bool done = false;
                for (uint32_t probes = 0; probes < DIST_SLOTS; probes++) {
                    const unsigned long long old = atomicCAS((unsigned long long *) &lk[off], (unsigned long long) DKEY_EMPTY,
                                                             (unsigned long long) key);
                    if (old == DKEY_EMPTY || old == key) { atomicAdd(&lc[off], 1u); done = true; break; }
                    off = (off + 1) & (DIST_SLOTS - 1);
                }
                if (!done) sh_flag = 1u;
            }
            __syncthreads();
            if (sh_flag) { // (S stays below 2^12: the selector has 12 bits, and 4096 sub-passes hold <= 1 distinct h2 slice each)
                if (S < 2048u && sp + 2 <= 32) { stk[sp++] = ((2u * S) << 16) | sub; stk[sp++] = ((2u * S) << 16) | (sub + S); }
                else if (tid == 0) atomicOr(err, DERR_TABLE_FULL); // never silently short
                __syncthreads();
                continue;
            }
            // append the occupied slots in slot order (ballot ranks; the wave totals go through LDS)
            for (uint32_t s0 = 0; s0 < DIST_SLOTS; s0 += DIST_THREADS) {
                const uint32_t s = s0 + tid;
                const bool occ = lk[s] != DKEY_EMPTY;
                const uint64_t bm = __ballot(occ);
                if (lane == 0) wtot[wave] = (uint32_t) __popcll(bm);
                __syncthreads();
                uint32_t pre = 0, tot = 0;
#pragma unroll
                for (int w = 0; w < DIST_THREADS / 64; w++) {
                    const uint32_t v = wtot[w];
                    pre += (uint32_t) w < wave ? v : 0u;
                    tot += v;
                }
                if (tid == 0) sh_base = tot ? atomicAdd(&nd[seq], tot) : 0u;
                __syncthreads();
                if (occ) {
                    const uint64_t at = b0 + sh_base + pre + (uint32_t) __popcll(bm & ((1ull << lane) - 1ull));
                    tmp_k[at] = lk[s];
                    tmp_c[at] = lc[s];
                }
                __syncthreads();
            }
            if (tid == 0 && sh_ones) {
                const uint64_t at = b0 + atomicAdd(&nd[seq], 1u);
                tmp_k[at] = DKEY_EMPTY;
                tmp_c[at] = sh_ones;
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

// tmp lists (indexed like the k-mers) -> dense output at dist_off[i]
__global__ void __launch_bounds__(256) k_dist_compact(const uint64_t *tmp_k, const uint32_t *tmp_c, const uint64_t *koff,
                                                      const uint64_t *dist_off, uint32_t n_seq, uint64_t *out_k, uint32_t *out_c) {
    for (uint32_t i = blockIdx.x; i < n_seq; i += gridDim.x) {
        const uint64_t src = koff[i], dst = dist_off[i], n = dist_off[i + 1] - dst;
        for (uint64_t j = threadIdx.x; j < n; j += blockDim.x) {
            out_k[dst + j] = tmp_k[src + j];
            out_c[dst + j] = tmp_c[src + j];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// ntHash
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rotr64(uint64_t x, unsigned r) {
    r &= 63u;
    return (x >> r) | (x << ((64u - r) & 63u));
}
template <bool T8>
__device__ __forceinline__ uint64_t nt_fwd(uint32_t code) { return T8 ? nt8b_fwd(code) : nt_seed(code); }
template <bool T8>
__device__ __forceinline__ uint64_t nt_rev(uint32_t code) { return T8 ? nt8b_rev(code) : nt_seed(3u - code); }

// inclusive XOR scan over the 64 lanes (DPP row shifts + row broadcasts, like wave_incl_scan_u32 with ^ for +)
__device__ __forceinline__ uint32_t wave_incl_xor_u32(uint32_t v) {
    v ^= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);
    v ^= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);
    v ^= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);
    v ^= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);
    v ^= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);
    v ^= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint64_t wave_incl_xor_u64(uint64_t v) {
    return ((uint64_t) wave_incl_xor_u32((uint32_t) (v >> 32)) << 32) | wave_incl_xor_u32((uint32_t) v);
}
__device__ __forceinline__ uint64_t shfl_down_u64(uint64_t v, int d) {
    return ((uint64_t) shfl_down_u32((uint32_t) (v >> 32), d) << 32) | shfl_down_u32((uint32_t) v, d);
}

struct NtArgs {
    const uint8_t *bases;
    const uint64_t *offsets;
    const uint64_t *packed_offsets;
    uint32_t n_seq;
    int packed;
    uint64_t total_bytes;
    int k;
    int mode;     // kmu_nthash_mode
    int n_hashes; // >= 1
    uint64_t *out;
    uint8_t *strand; // may be null
    uint32_t *err;
    const uint16_t *novalid; // k_nthash_flat: bit p of word w set = position 16 w + p starts no k-mer (flat_novalid, kmu_smer.hpp)
};

// A wave step covers 64 code words but emits the k-mers that start in the first 62 (the window of a k-mer, k <= 32, ends at
// most two words further on): the steps of a sequence advance by 62 words and no halo has to be fetched separately.
template <bool T8>
__global__ void __launch_bounds__(256) k_nthash(NtArgs a) {
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, lane = lane_id();
    const int k = a.k;
    const uint32_t m = (uint32_t) k & 15u;
    const int d = k >> 4;
    const uint64_t mult = (uint64_t) k * 0x90b45d39fb6da1faull; // ksize * MULTISEED, nthash.rs:68
    for (uint32_t i = blockIdx.x; i < a.n_seq; i += gridDim.x) {
        SeqView s;
        s.base = a.bases;
        s.len = a.offsets[i + 1] - a.offsets[i];
        s.packed = a.packed;
        if (a.packed) {
            s.begin = a.packed_offsets[i];
            s.total = a.total_bytes ? a.total_bytes
                                    : (a.packed_offsets[a.n_seq - 1] + (a.offsets[a.n_seq] - a.offsets[a.n_seq - 1] + 3) / 4);
        } else {
            s.begin = a.offsets[i];
            s.total = a.total_bytes ? a.total_bytes : a.offsets[a.n_seq];
        }
        const uint64_t L = s.len, nk = L >= (uint64_t) k ? L - k + 1 : 0;
        uint32_t bad = 0;
        if (nk == 0) {
            bad |= wave_validate_seq(s, wave, nwaves, false);
        } else {
            const uint32_t lead = seq_lead(s);
            const uint64_t nwords = seq_num_words(s), nsteps = (nwords + 61) / 62;
            uint64_t *o = a.out + a.offsets[i] * (uint64_t) a.n_hashes;
            uint8_t *so = a.strand ? a.strand + a.offsets[i] : nullptr;
            for (uint64_t st = wave; st < nsteps; st += nwaves) {
                const uint64_t widx = st * 62 + (uint64_t) lane;
                uint32_t b;
                const uint32_t w0 = load_code_word(s, widx, b);
                bad |= b;
                const uint32_t w1 = shfl_down_u32(w0, 1), w2 = shfl_down_u32(w0, 2);
                // XOR-prefixes of the rotated seeds: forward seeds turned right by the base's index t in the step, complement
                // seeds turned left by it; xf / xr = the prefix up to (not including) base m of this lane
                uint64_t pf = 0, pr = 0, xf = 0, xr = 0;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const uint32_t code = (w0 >> (30 - 2 * j)) & 3u;
                    const unsigned t = (unsigned) (16 * lane + j) & 63u;
                    if ((uint32_t) j == m) { xf = pf; xr = pr; }
                    pf ^= rotr64(nt_fwd<T8>(code), t);
                    pr ^= rotl64(nt_rev<T8>(code), t);
                }
                const uint64_t ef = wave_incl_xor_u64(pf) ^ pf, er = wave_incl_xor_u64(pr) ^ pr; // prefix at this lane's base 0
                const uint64_t kf = shfl_down_u64(ef ^ xf, d), kr = shfl_down_u64(er ^ xr, d);   // prefix at base 0 + k
                uint64_t F = rotl64(kf ^ ef, (unsigned) (16 * lane + k - 1) & 63u);
                uint64_t R = rotr64(kr ^ er, (unsigned) (16 * lane) & 63u);
                const uint64_t hi = ((uint64_t) w0 << 32) | w1;
                const int64_t p0 = (int64_t) (widx * 16) - (int64_t) lead;
                const int sh = 64 - 2 * k;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const uint64_t v = (hi << (2 * j)) | (((uint64_t) w2 << (2 * j)) >> 32);
                    const uint64_t val = v >> sh;
                    if (j > 0) { // roll from position j - 1: its first base leaves, the last base of this k-mer enters
                        const uint32_t oldb = (w0 >> (30 - 2 * (j - 1))) & 3u, newb = (uint32_t) val & 3u;
                        F = rotl64(F, 1) ^ rotl64(nt_fwd<T8>(oldb), (unsigned) k) ^ nt_fwd<T8>(newb);
                        R = rotr64(R, 1) ^ rotr64(nt_rev<T8>(oldb), 1) ^ rotl64(nt_rev<T8>(newb), (unsigned) (k - 1));
                    }
                    const int64_t p = p0 + j;
                    if (lane < 62 && p >= 0 && p < (int64_t) nk) {
                        uint64_t h0;
                        uint8_t sd;
                        if (a.mode == KMU_NTHASH_FORWARD) { h0 = F; sd = 0; }
                        else if (a.mode == KMU_NTHASH_RCOMP) { h0 = R; sd = 1; }
                        else if (F <= R) { h0 = F; sd = 0; } // nthash.rs:223-227
                        else { h0 = R; sd = 1; }
                        uint64_t *op = o + (uint64_t) p * (uint64_t) a.n_hashes;
                        op[0] = h0;
                        for (int q = 1; q < a.n_hashes; q++) { // from_one_hash_val_to_mult_hash, nthash.rs:63-72
                            uint64_t t = h0 * ((uint64_t) q ^ mult);
                            t ^= t >> 27;
                            op[q] = t;
                        }
                        if (so) so[p] = sd;
                    }
                }
            }
        }
        if (bad) atomicOr(a.err, DERR_NON_ACGT);
    }
}

// The same over the FLAT base stream of unpacked reads (the walk of the count build): a wave step is 62 words of the whole
// array wherever the reads begin and end -- 150 bp reads fill the lanes like 100 kb reads do -- and a k-mer is emitted
// when it lies inside one read.  The hash of a k-mer only involves its own k bases (the prefixes cancel outside), so
// nothing has to be restarted at a read boundary; out is indexed by the absolute base position.
// ONE (n_hashes == 1): a lane's sixteen hashes lie 128 bytes apart in `out`, and stored from the registers every store of a wave is
// 64 requests of 8 bytes; they go through LDS (lane rows padded to 17 words: no bank is hit twice by a quarter-wave) and leave as
// 512 contiguous bytes per instruction.  A position without a k-mer is not written, as before.
template <bool T8, bool ONE>
__global__ void __launch_bounds__(256) k_nthash_flat(NtArgs a) {
    __shared__ uint64_t stage_all[ONE ? 4 : 1][ONE ? 64 * 17 : 1];
    __shared__ uint32_t nvs_all[ONE ? 4 : 1][ONE ? 64 : 1];
    uint64_t *stage = stage_all[ONE ? threadIdx.x >> 6 : 0];
    uint32_t *nvs = nvs_all[ONE ? threadIdx.x >> 6 : 0];
    const int lane = lane_id();
    const int k = a.k;
    const uint32_t m = (uint32_t) k & 15u;
    const int d = k >> 4;
    const uint64_t mult = (uint64_t) k * 0x90b45d39fb6da1faull;
    const uint64_t total = a.offsets[a.n_seq], start = a.offsets[0];
    SeqView s;
    s.base = a.bases; s.begin = 0; s.len = total; s.total = total; s.packed = 0;
    const uint64_t w_first = start / 16, nwords = (total + 15) / 16 - w_first, nsteps = (nwords + 61) / 62;
    const uint64_t wave_global = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves_global = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    uint32_t bad = 0;
    for (uint64_t st = wave_global; st < nsteps; st += nwaves_global) {
        const uint64_t widx = w_first + st * 62 + (uint64_t) lane;
        uint32_t b;
        const uint32_t w0 = load_code_word(s, widx, b);
        const uint64_t g0 = widx * 16;
        if (g0 < start) b &= ~((1u << (start - g0 > 16 ? 16 : (uint32_t) (start - g0))) - 1u); // bytes before the first read
        bad |= b;
        const uint32_t w1 = shfl_down_u32(w0, 1), w2 = shfl_down_u32(w0, 2);
        uint64_t pf = 0, pr = 0, xf = 0, xr = 0;
#pragma unroll 4
        for (int j = 0; j < 16; j++) {
            const uint32_t code = (w0 >> (30 - 2 * j)) & 3u;
            const unsigned t = (unsigned) (16 * lane + j) & 63u;
            if ((uint32_t) j == m) { xf = pf; xr = pr; }
            pf ^= rotr64(nt_fwd<T8>(code), t);
            pr ^= rotl64(nt_rev<T8>(code), t);
        }
        const uint64_t ef = wave_incl_xor_u64(pf) ^ pf, er = wave_incl_xor_u64(pr) ^ pr;
        const uint64_t kf = shfl_down_u64(ef ^ xf, d), kr = shfl_down_u64(er ^ xr, d);
        uint64_t F = rotl64(kf ^ ef, (unsigned) (16 * lane + k - 1) & 63u);
        uint64_t R = rotr64(kr ^ er, (unsigned) (16 * lane) & 63u);
        // which of the lane's positions start a k-mer is looked up (one 16-bit load next to the lane's bases), not derived from the
        // read offsets: a wave-wide search of the lane's read and a dependent look-up at every read boundary made a step a chain of
        // ~20 loads (config 2's 150 bp reads: 1.16 ms for 150 M positions, the waves waiting for them whatever their number)
        const bool in = g0 < total && g0 + 16 > start && lane < 62;
        const uint32_t nv = in ? (uint32_t) a.novalid[widx] : 0xFFFFu;
        const uint64_t hi = ((uint64_t) w0 << 32) | w1;
        const int sh = 64 - 2 * k;
        uint32_t sdm = 0; // ONE: bit j = position j's hash is the reverse strand's
#pragma unroll 4
        for (int j = 0; j < 16; j++) {
            const uint64_t v = (hi << (2 * j)) | (((uint64_t) w2 << (2 * j)) >> 32);
            const uint64_t val = v >> sh;
            if (j > 0) {
                const uint32_t oldb = (w0 >> (30 - 2 * (j - 1))) & 3u, newb = (uint32_t) val & 3u;
                F = rotl64(F, 1) ^ rotl64(nt_fwd<T8>(oldb), (unsigned) k) ^ nt_fwd<T8>(newb);
                R = rotr64(R, 1) ^ rotr64(nt_rev<T8>(oldb), 1) ^ rotl64(nt_rev<T8>(newb), (unsigned) (k - 1));
            }
            uint64_t h0;
            uint32_t sd;
            if (a.mode == KMU_NTHASH_FORWARD) { h0 = F; sd = 0; }
            else if (a.mode == KMU_NTHASH_RCOMP) { h0 = R; sd = 1; }
            else if (F <= R) { h0 = F; sd = 0; }
            else { h0 = R; sd = 1; }
            if (ONE) {
                stage[17 * lane + j] = h0;
                sdm |= sd << j;
            } else if (!((nv >> j) & 1u)) {
                const uint64_t g = g0 + j;
                uint64_t *op = a.out + g * (uint64_t) a.n_hashes;
                op[0] = h0;
                for (int q = 1; q < a.n_hashes; q++) {
                    uint64_t t = h0 * ((uint64_t) q ^ mult);
                    t ^= t >> 27;
                    op[q] = t;
                }
                if (a.strand) a.strand[g] = (uint8_t) sd;
            }
        }
        if (ONE) { // (the wave's own rows: its LDS requests are served in order)
            nvs[lane] = nv | (sdm << 16);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint64_t gbase = (w_first + st * 62) * 16; // position of lane 0's first base
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t q = (uint32_t) i * 64u + (uint32_t) lane, src = q >> 4, jj = q & 15u;
                const uint32_t mk = nvs[src];
                if (!((mk >> jj) & 1u)) {
                    a.out[gbase + q] = stage[17u * src + jj];
                    if (a.strand) a.strand[gbase + q] = (uint8_t) ((mk >> (16u + jj)) & 1u);
                }
            }
            __builtin_amdgcn_wave_barrier(); // (the next step's rows overwrite these)
        }
    }
    if (bad) atomicOr(a.err, DERR_NON_ACGT);
}

} // namespace kmu

using namespace kmu;

extern "C" int kmu_nthash(kmu_ctx *ctx, const kmu_nthash_params *p, const uint8_t *bases, const uint64_t *offsets,
                          const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *hashes_out, uint8_t *strand_out) {
    if (!ctx || !p || !hashes_out) return KMU_E_BAD_ARG;
    if (p->kmer_size < 1 || p->kmer_size > 32) return fail(ctx, KMU_E_BAD_K, "kmu_nthash: 1 <= kmer_size <= 32 (got %d)", p->kmer_size);
    if (p->n_hashes < 1 || p->n_hashes > 1024) return fail(ctx, KMU_E_BAD_ARG, "kmu_nthash: 1 <= n_hashes <= 1024");
    if (p->mode < KMU_NTHASH_CANONICAL || p->mode > KMU_NTHASH_RCOMP) return fail(ctx, KMU_E_BAD_ARG, "kmu_nthash: bad mode %d", p->mode);
    if (p->table != KMU_NTHASH_TABLE_2B && p->table != KMU_NTHASH_TABLE_8B) return fail(ctx, KMU_E_BAD_ARG, "kmu_nthash: bad table %d", p->table);
    if (p->input_kind == KMU_INPUT_PACKED2 && p->table == KMU_NTHASH_TABLE_8B)
        return fail(ctx, KMU_E_BAD_ARG, "the 8-bit ntHash table is defined on ASCII bases (nthash.rs:48-57)");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, packed_offsets, n_seq, p->input_kind, p->mem, &ds));
    uint64_t *d_out = hashes_out;
    uint8_t *d_strand = strand_out;
    uint64_t total = 0;
    const uint64_t off0 = p->mem == KMU_MEM_HOST && n_seq ? offsets[0] : 0;
    if (p->mem == KMU_MEM_HOST) {
        total = n_seq ? offsets[n_seq] - off0 : 0;
        void *q;
        KMU_TRY(dev_buf(ctx, "out.u64", (size_t) total * 8 * p->n_hashes + 8, &q));
        d_out = (uint64_t *) q;
        KMU_HIP(ctx, hipMemsetAsync(d_out, 0, (size_t) total * 8 * p->n_hashes, ctx->stream));
        if (strand_out) {
            KMU_TRY(dev_buf(ctx, "out.u8", (size_t) total + 8, &q));
            d_strand = (uint8_t *) q;
            KMU_HIP(ctx, hipMemsetAsync(d_strand, 0, (size_t) total, ctx->stream));
        }
    }
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    if (n_seq) {
        NtArgs a{ds.bases, ds.offsets, ds.packed_offsets, n_seq, ds.packed, ds.total_bytes, p->kmer_size, p->mode, p->n_hashes,
                 d_out, d_strand, d_err, nullptr};
        if (!ds.packed) { // the positions that start no k-mer, once per call (the end of the stream as the kernel sees it: offsets[n_seq])
            uint64_t end = 0;
            KMU_HIP(ctx, hipMemcpyAsync(&end, ds.offsets + n_seq, 8, hipMemcpyDeviceToHost, ctx->stream));
            KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
            void *nvp;
            KMU_TRY(flat_novalid(ctx, ds, end, p->kmer_size, (end + 15) / 16 + 128, "nt.novalid", &nvp));
            a.novalid = (const uint16_t *) nvp;
        }
        KernelTimer t(ctx, "k_nthash");
        if (!ds.packed) { // one flat stream: the lanes are full whatever the read lengths
            const int grid = ctx->num_cus * 8;
            const bool t8 = p->table == KMU_NTHASH_TABLE_8B, one = p->n_hashes == 1;
            const auto kern = t8 ? (one ? k_nthash_flat<true, true> : k_nthash_flat<true, false>) : (one ? k_nthash_flat<false, true> : k_nthash_flat<false, false>);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, ctx->stream, a);
        } else {
            const int grid = (int) std::min<uint32_t>(n_seq, (uint32_t) ctx->num_cus * 8);
            hipLaunchKernelGGL(k_nthash<false>, dim3(grid), dim3(256), 0, ctx->stream, a);
        }
    }
    KMU_HIP(ctx, hipGetLastError());
    if (p->mem == KMU_MEM_HOST) {
        KMU_HIP(ctx, hipMemcpyAsync(hashes_out + off0 * p->n_hashes, d_out, (size_t) total * 8 * p->n_hashes, hipMemcpyDeviceToHost, ctx->stream));
        if (strand_out) KMU_HIP(ctx, hipMemcpyAsync(strand_out + off0, d_strand, (size_t) total, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!(p->mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
    return finish_call(ctx, p->mem);
}

extern "C" int kmu_kmer_distribution(kmu_ctx *ctx, const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                                     const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *kmers_out, uint32_t *mult_out,
                                     uint64_t cap, uint64_t *dist_offsets_out, uint64_t *n_out) {
    if (!ctx || !p || !n_out || (kmers_out && !mult_out)) return KMU_E_BAD_ARG;
    KMU_TRY(check_kmer(ctx, p->kmer_type, p->kmer_size));
    if (!fhash_valid(p->fhash, p->kmer_type)) return fail(ctx, KMU_E_BAD_ARG, "fhash %d not valid for kmer_type %d", p->fhash, p->kmer_type);
    if (p->input_kind == KMU_INPUT_PACKED2 && (kmer_is_aa(p->kmer_type) || p->fhash == KMU_FHASH_CANON_NTHASH_8B))
        return fail(ctx, KMU_E_BAD_ARG, "packed input not valid for this kmer_type / fhash");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    *n_out = 0;
    if (n_seq == 0) {
        if (dist_offsets_out && p->mem == KMU_MEM_HOST) dist_offsets_out[0] = 0;
        return KMU_OK;
    }
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, packed_offsets, n_seq, p->input_kind, p->mem, &ds));
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    void *koff, *vals, *passes, *item_off, *nd, *tk, *tc, *doff, *queue;
    KMU_TRY(dev_buf(ctx, "all.koff", ((size_t) n_seq + 1) * 8, &koff));
    hipLaunchKernelGGL(k_nk_scan, dim3(1), dim3(1024), 0, ctx->stream, ds.offsets, n_seq, p->kmer_size, (uint64_t *) koff, d_err);
    uint64_t n_kmers = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&n_kmers, (uint64_t *) koff + n_seq, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    KMU_TRY(dev_buf(ctx, "all.hashes", n_kmers * 8 + 64, &vals));
    {
        KmerCfg cfg{p->kmer_type, p->kmer_size, p->fhash};
        const int spread = n_seq < (uint32_t) ctx->num_cus * 4 ? 1 : 0;
        const int grid = spread ? ctx->num_cus * 8 : (int) std::min<uint32_t>(n_seq, (uint32_t) ctx->num_cus * 8);
        KernelTimer t(ctx, "k_seq_hashes_compact");
        hipLaunchKernelGGL(k_seq_hashes_compact, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets, ds.packed_offsets,
                           n_seq, ds.packed, ds.total_bytes, cfg, (const uint64_t *) koff, (uint64_t *) vals, d_err, spread);
    }
    KMU_TRY(dev_buf(ctx, "dist.passes", ((size_t) n_seq + 1) * 4, &passes));
    KMU_TRY(dev_buf(ctx, "dist.item_off", ((size_t) n_seq + 1) * 8, &item_off));
    KMU_TRY(dev_buf(ctx, "dist.nd", ((size_t) n_seq + 1) * 4, &nd));
    KMU_TRY(dev_buf(ctx, "dist.doff", ((size_t) n_seq + 1) * 8, &doff));
    KMU_TRY(dev_buf(ctx, "dist.tmp_k", n_kmers * 8 + 64, &tk));
    KMU_TRY(dev_buf(ctx, "dist.tmp_c", n_kmers * 4 + 64, &tc));
    KMU_TRY(dev_buf(ctx, "dist.queue", 64, &queue));
    KMU_HIP(ctx, hipMemsetAsync(nd, 0, ((size_t) n_seq + 1) * 4, ctx->stream));
    KMU_HIP(ctx, hipMemsetAsync(queue, 0, 64, ctx->stream));
    const int grid_s = (int) std::min<uint64_t>(((uint64_t) n_seq + 255) / 256, (uint64_t) ctx->num_cus * 8);
    hipLaunchKernelGGL(k_dist_plan, dim3(grid_s), dim3(256), 0, ctx->stream, (const uint64_t *) koff, n_seq, (uint32_t *) passes);
    KMU_TRY(device_scan_u32(ctx, (const uint32_t *) passes, n_seq, (uint64_t *) item_off));
    uint64_t n_items = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&n_items, (uint64_t *) item_off + n_seq, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_items) {
        const int grid = (int) std::min<uint64_t>(n_items, (uint64_t) ctx->num_cus * 3);
        KernelTimer t(ctx, "k_kmer_dist");
        hipLaunchKernelGGL(k_kmer_dist, dim3(grid), dim3(DIST_THREADS), 0, ctx->stream, (const uint64_t *) vals, (const uint64_t *) koff,
                           (const uint64_t *) item_off, n_seq, n_items, (uint64_t *) tk, (uint32_t *) tc, (uint32_t *) nd,
                           (unsigned long long *) queue, d_err);
    }
    KMU_TRY(device_scan_u32(ctx, (const uint32_t *) nd, n_seq, (uint64_t *) doff));
    KMU_HIP(ctx, hipGetLastError());
    uint64_t n_pairs = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&n_pairs, (uint64_t *) doff + n_seq, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_TRY(check_err_word(ctx, d_err)); // (synchronises)
    *n_out = n_pairs;
    if (dist_offsets_out) {
        if (p->mem == KMU_MEM_HOST) KMU_HIP(ctx, hipMemcpy(dist_offsets_out, doff, ((size_t) n_seq + 1) * 8, hipMemcpyDeviceToHost));
        else KMU_HIP(ctx, hipMemcpyAsync(dist_offsets_out, doff, ((size_t) n_seq + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (!kmers_out) return finish_call(ctx, p->mem); // sizes only
    if (cap < n_pairs) return fail(ctx, KMU_E_BAD_ARG, "output too small: %llu pairs", (unsigned long long) n_pairs);
    uint64_t *d_k = kmers_out;
    uint32_t *d_c = mult_out;
    if (p->mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "dist.out_k", n_pairs * 8 + 64, &q));
        d_k = (uint64_t *) q;
        KMU_TRY(dev_buf(ctx, "dist.out_c", n_pairs * 4 + 64, &q));
        d_c = (uint32_t *) q;
    }
    if (n_pairs) {
        const int grid = (int) std::min<uint32_t>(n_seq, (uint32_t) ctx->num_cus * 8);
        hipLaunchKernelGGL(k_dist_compact, dim3(grid), dim3(256), 0, ctx->stream, (const uint64_t *) tk, (const uint32_t *) tc,
                           (const uint64_t *) koff, (const uint64_t *) doff, n_seq, d_k, d_c);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (p->mem == KMU_MEM_HOST) {
        KMU_HIP(ctx, hipMemcpyAsync(kmers_out, d_k, n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipMemcpyAsync(mult_out, d_c, n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    return finish_call(ctx, p->mem);
}
