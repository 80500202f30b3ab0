// kmu_smer.hip -- the sender's half of a distributed count with minimizer owners: the k-mers of a rank's reads grouped by
// owner as SUPER-K-MER RECORDS (kmu_smer.h), without ever forming a k-mer.
//
// Reference: the producer of count_kmer_threaded_one_to_many (src/base/kmercount.rs:933-949) generates every k-mer,
// canonicalises it and sends it to `dispatch(n)`; what is dispatched is one k-mer per message (:936-943).  Here the
// "message" to an owner is a run of up to 16 consecutive k-mers of a read that share the owner, as 2-bit bases.
//
// Two kernels over the flat base stream (one aligned 16-byte word of ASCII per lane, like every kernel of the path):
//   k_smer_census   records and k-mers per (unit, owner), validation of the bases, the duplication sample;
//   k_smer_scatter  the records, to exact private ranges (unit, owner) computed from the census -- no global atomics.
// A wave step covers 64 code words and emits the k-mers of the first 61: the minimizer window of a k-mer reaches 2 words
// beyond its own, the bases of a record 3, so nothing is fetched twice and no halo load exists.  Per lane: the hashes of
// the 16 canonical m-mers that start in its word, prefix / suffix minima, the neighbours' prefix minima by DPP wave shifts
// (no LDS), hence the minimizer hash -- and the owner -- of its 16 k-mers in ~25 instructions per k-mer; runs, record
// starts and lengths are bit-mask arithmetic on 16-bit masks with one max-scan across the wave.
#include <algorithm>
#include <type_traits>

#include "kmu_flat.h"
#include "kmu_smer.hpp"

namespace kmu {

static constexpr int SMER_STEP_WORDS = 61; // code words whose k-mers one wave step emits
static constexpr int SMER_THREADS = 256;
static constexpr uint32_t SMER_FLUSH_STEPS = 1024; // census: a lane's 16-bit fields take 31 per step

// lane i <- lane i + 1 (lane 63 <- 0) / lane i <- lane i - 1 (lane 0 <- 0): full-rate DPP moves, no LDS crossbar
__device__ __forceinline__ uint32_t dpp_shl1(uint32_t v) { return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t dpp_shr1(uint32_t v) { return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
// inclusive prefix maximum over the 64 lanes (identity 0), the max form of wave_incl_scan_u32
__device__ __forceinline__ uint32_t wave_incl_scan_max_u32(uint32_t v) {
    v = umax32(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false));
    v = umax32(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false));
    v = umax32(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false));
    v = umax32(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false));
    v = umax32(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false));
    v = umax32(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false));
    return v;
}

// minimizer hash of the 16 k-mers that start in this lane's word: mw[j] = min over the W canonical m-mers at offsets
// j .. j + W - 1 from the lane's first base.  h[i] = hash of the m-mer at offset i (own word: i < 16; they reach base 29 at
// most: w0 and w1 suffice); the windows continue in the next lane's h (offsets 16 .. 31) and the one after (32 .. 35).
template <int W>
__device__ __forceinline__ void smer_minwin(uint32_t w0, uint32_t w1, int m, uint32_t (&mw)[16]) {
    const uint64_t hi = ((uint64_t) w0 << 32) | w1;
    const uint32_t mask = (1u << (2 * m)) - 1u;
    const int sh0 = 64 - 2 * m;
    // (smer_hash of every m-mer, with the reverse complements taken out of the reverse complement of the two words: the m-mer at
    //  offset i ends 2 i bits above its bottom -- StepWin's form, kmu_device.h)
    const uint64_t R = ((uint64_t) revcomp16(w1) << 32) | revcomp16(w0);
    uint32_t h[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t f = (uint32_t) (hi >> (sh0 - 2 * i)) & mask, r = (uint32_t) (R >> (2 * i)) & mask;
        h[i] = smer_mix(r < f ? r : f);
    }
    uint32_t P[16], S[16]; // prefix / suffix minima of the own word
    P[0] = h[0];
#pragma unroll
    for (int i = 1; i < 16; i++) P[i] = umin32(P[i - 1], h[i]);
    S[15] = h[15];
#pragma unroll
    for (int i = 14; i >= 0; i--) S[i] = umin32(S[i + 1], h[i]);
    if (W == 21) { // offsets j .. 15 | 16 .. min(31, j + 20) | 32 .. j + 20 (j >= 12)
        uint32_t P1[16], P2[4];
#pragma unroll
        for (int t = 4; t < 16; t++) P1[t] = dpp_shl1(P[t]);
#pragma unroll
        for (int t = 0; t < 4; t++) P2[t] = dpp_shl1(dpp_shl1(P[t]));
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t a = umin32(S[j], P1[j + 4 < 15 ? j + 4 : 15]);
            if (j >= 12) a = umin32(a, P2[j - 12]);
            mw[j] = a;
        }
    } else if (W == 16) { // offsets j .. 15 | 16 .. j + 15 (j >= 1)
#pragma unroll
        for (int j = 0; j < 16; j++) mw[j] = j ? umin32(S[j], dpp_shl1(P[j - 1])) : S[0];
    } else { // W == 9, blocks of eight: j <= 7: offsets j .. 7 | 8 .. j + 8;  j >= 8: offsets j .. 15 | 16 .. j + 8
        uint32_t SA[8], PB[8];
        SA[7] = h[7];
#pragma unroll
        for (int i = 6; i >= 0; i--) SA[i] = umin32(SA[i + 1], h[i]);
        PB[0] = h[8];
#pragma unroll
        for (int i = 1; i < 8; i++) PB[i] = umin32(PB[i - 1], h[8 + i]);
#pragma unroll
        for (int j = 0; j < 8; j++) mw[j] = umin32(SA[j], PB[j]);
#pragma unroll
        for (int j = 8; j < 16; j++) mw[j] = umin32(S[j], dpp_shl1(P[j - 8]));
    }
}

// One wave step, analysed: the lane's words, which of its 16 positions start a record, where runs break.
struct SmerLane {
    uint32_t w0, w1, w2, w3;
    uint32_t V;   // positions that start a k-mer of some read
    uint32_t K;   // positions a run cannot continue through: a boundary (owner change / first k-mer of a read) or no k-mer
    uint32_t rm;  // positions that start a record (output lanes only)
    uint32_t fbn; // the next lane's first break (0 .. 16); 0 behind the last output lane: runs end with the step
    uint32_t bad; // non-ACGT bytes of the lane's own word
};

// record length at start position s of a lane
__device__ __forceinline__ uint32_t smer_len(const SmerLane &a, uint32_t s) {
    const uint32_t kb = a.K & ~((2u << s) - 1u);
    const uint32_t L = kb ? (uint32_t) __builtin_ctz(kb) - s : 16u - s + a.fbn;
    return L < SMER_REC_KMERS ? L : SMER_REC_KMERS;
}

// Which positions of the flat stream start a k-mer is looked up, not derived: bit p of `novalid` is set where position p starts
// none -- the last k - 1 positions of every read, what lies before the first read and behind the last (k_smer_novalid, one
// thread per read, once per batch).  A lane's 16 positions are one 16-bit load next to its 16 bases; no kernel of this file
// searches the read offsets (round 3's count levels spend a wave-wide search and up to three dependent look-ups per step there).
__global__ void __launch_bounds__(256) k_smer_novalid(const uint64_t *offsets, uint32_t n_seq, int k, uint64_t total, uint64_t nbits, uint32_t *mask) {
    const uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r > (uint64_t) n_seq + 1) return;
    uint64_t lo, hi;
    if (r < n_seq) { // the positions of read r that start no k-mer
        const uint64_t b = offsets[r], e = offsets[r + 1];
        lo = e - b >= (uint64_t) k ? e - (uint64_t) (k - 1) : b;
        hi = e;
    } else if (r == n_seq) { lo = 0; hi = offsets[0]; }  // before the first read (a range of a larger read set)
    else { lo = total; hi = nbits; }                      // behind the last read, to the end of the last wave step's halo
    while (lo < hi) { // (reads: at most two words; the tail: a few dozen)
        const uint64_t w = lo >> 5, e = (w + 1) << 5 < hi ? (w + 1) << 5 : hi;
        const uint32_t n = (uint32_t) (e - lo), sh = (uint32_t) (lo & 31u);
        atomicOr(&mask[w], (n == 32u ? 0xFFFFFFFFu : ((1u << n) - 1u)) << sh);
        lo = e;
    }
}

// what a wave step needs from memory, requested a step ahead: the lane's 16 bases and its 16 "no k-mer" bits.  Unconditional
// loads from clamped addresses: their number in flight is a constant for the compiler's waits (see flat_step_fetch, kmu_count.hip).
struct SmerRaw {
    uint4 c;
    uint32_t nv;
};
__device__ __forceinline__ void smer_fetch(const uint8_t *bases, const uint16_t *novalid, uint64_t total, uint64_t nwords_mask, uint64_t st, SmerRaw &r) {
    const uint64_t widx = st * SMER_STEP_WORDS + (uint64_t) lane_id();
    r.c = make_uint4(0u, 0u, 0u, 0u);
    if (total >= 16) { // (wave-uniform)
        const uint64_t lastc = (total - 16) & ~15ull, a0 = widx * 16;
        r.c = *reinterpret_cast<const uint4 *>(bases + (a0 < lastc ? a0 : lastc));
    }
    r.nv = novalid[widx < nwords_mask ? widx : nwords_mask - 1];
}

// `own[j]`: owner of position j.
template <int W>
__device__ __forceinline__ void smer_step(const uint8_t *bases, uint64_t total, const SmerCfg &cf, uint32_t n_parts, uint64_t st,
                                          const SmerRaw &raw, SmerLane &a, uint32_t (&mw)[16], uint32_t (&own)[16]) {
    const uint32_t lane = (uint32_t) lane_id();
    const uint64_t widx = st * SMER_STEP_WORDS + lane;
    if (__all(widx * 16 + 16 <= total)) a.w0 = pack16_ascii(raw.c, a.bad); // every chunk of the step whole: all but the stream's last step
    else {
        SeqView s;
        s.base = bases; s.begin = 0; s.len = total; s.total = total; s.packed = 0;
        a.w0 = load_code_word(s, widx, a.bad);
    }
    a.w1 = dpp_shl1(a.w0);
    a.w2 = dpp_shl1(a.w1);
    a.w3 = dpp_shl1(a.w2);
    smer_minwin<W>(a.w0, a.w1, cf.m, mw);
    // own[j] = smer_owner_of(mw[j], n_parts); 2 / 4 / 8 ranks: the remainder is a mask (a division per k-mer would dominate the
    // kernel), decided once per step, not per k-mer
    if ((n_parts & (n_parts - 1u)) == 0u) {
        const uint32_t nm = n_parts - 1u;
#pragma unroll
        for (int j = 0; j < 16; j++) own[j] = ((mw[j] * 0xC2B2AE35u) >> 16) & nm;
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) own[j] = ((mw[j] * 0xC2B2AE35u) >> 16) % n_parts;
    }
    const uint32_t V = ~raw.nv & 0xFFFFu;
    // ---- boundaries: the first k-mer behind a position without one (every read's first k-mer: the k - 1 positions before it
    // start none), an owner change, and the step's first position ----
    const uint32_t pv = dpp_shr1(V) >> 15, po = dpp_shr1(own[15]);
    uint32_t chg = own[0] != po ? 1u : 0u;
#pragma unroll
    for (int j = 1; j < 16; j++) chg |= (own[j] != own[j - 1] ? 1u : 0u) << j;
    const uint32_t Vp = ((V << 1) | (lane ? pv : 0u)) & 0xFFFFu;
    const uint32_t B = V & (~Vp | chg);
    a.V = V;
    a.K = (B | ~V) & 0xFFFFu;
    // ---- record starts: every boundary, and every 16th position of a run: the run that enters this lane started at the last
    // boundary before it (between a boundary and a later position of its run lies no position without a k-mer) ----
    const uint32_t e = B ? 16u * lane + (31u - (uint32_t) __builtin_clz(B)) + 1u : 0u;
    const uint32_t excl = dpp_shr1(wave_incl_scan_max_u32(e));
    uint32_t rm = B;
    if ((V & 1u) && !(B & 1u)) { // (excl != 0: lane 0 starts with a boundary)
        const uint32_t d = 16u * lane - (excl - 1u), cj = (16u - (d & 15u)) & 15u;
        if ((a.K & ((2u << cj) - 1u)) == 0u) rm |= 1u << cj;
    }
    const uint32_t fb = a.K ? (uint32_t) __builtin_ctz(a.K) : 16u;
    const uint32_t fb_next = dpp_shl1(fb); // (a DPP move reads nothing from a lane that is switched off: never under a lane-dependent condition)
    a.fbn = lane + 1 < (uint32_t) SMER_STEP_WORDS ? fb_next : 0u;
    a.rm = lane < (uint32_t) SMER_STEP_WORDS ? rm : 0u;
    if (lane >= (uint32_t) SMER_STEP_WORDS) a.bad = 0; // (a halo word is some other step's own word)
}

// the fields of two packed per-owner counters (eight 8-bit fields each: owners 0 .. 7) spread to 16-bit fields:
// word i holds owner (i >> 1) + 4 (i & 1) in its low half and that + 2 in its high half
__device__ __forceinline__ void spread8(uint64_t p, uint32_t (&w)[4]) {
    const uint64_t ev = p & 0x00FF00FF00FF00FFull, od = (p >> 8) & 0x00FF00FF00FF00FFull;
    w[0] = (uint32_t) ev; w[1] = (uint32_t) (ev >> 32); w[2] = (uint32_t) od; w[3] = (uint32_t) (od >> 32);
}
__device__ __forceinline__ uint32_t field16(const uint32_t (&w)[4], uint32_t o) { // owner o's field of spread words
    const uint32_t i = ((o & 1u) << 1) | ((o >> 2) & 1u);
    const uint32_t x = i == 0u ? w[0] : i == 1u ? w[1] : i == 2u ? w[2] : w[3];
    return (o & 2u) ? x >> 16 : x & 0xFFFFu;
}
__device__ __forceinline__ uint32_t sel16(const uint32_t (&v)[16], uint32_t s) {
    uint32_t r = v[0];
#pragma unroll
    for (int j = 1; j < 16; j++) r = s == (uint32_t) j ? v[j] : r;
    return r;
}
// the owner of position s: up to eight owners travel as sixteen nibbles of two registers (one shift instead of fifteen selects)
template <bool PACK8>
struct OwnerOf {
    uint32_t lo = 0, hi = 0;
    __device__ __forceinline__ void set(const uint32_t (&own)[16]) {
        if (!PACK8) return;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            lo |= own[j] << (4 * j);
            hi |= own[8 + j] << (4 * j);
        }
    }
    __device__ __forceinline__ uint32_t at(const uint32_t (&own)[16], uint32_t s) const {
        if (!PACK8) return sel16(own, s);
        return ((s & 8u ? hi : lo) >> (4u * (s & 7u))) & 15u;
    }
};

struct SmerGeom {
    uint64_t total, start, nsteps;
    uint32_t steps_per_unit;
};

// PACK8 (n_parts <= 8: one node's GPUs): per-owner counts travel in 8-bit fields of a register and reach LDS as wave sums;
// otherwise one LDS atomic per record.
template <int W, bool PACK8>
__global__ void __launch_bounds__(SMER_THREADS) k_smer_census(const uint8_t *bases, const uint16_t *novalid, uint64_t total, int k,
                                                               uint32_t n_parts, uint32_t steps_per_unit, uint32_t *hist,
                                                               unsigned long long *kmers_per_owner, uint32_t *err, SampleArgs sa) {
    extern __shared__ uint32_t lh[]; // [n_parts] records, [n_parts] k-mers, then the sample: counter (2 words) + list
    uint32_t *lk = lh + n_parts;
    uint32_t *ls_n = lk + n_parts + ((2u * n_parts) & 1u); // 8-byte aligned
    uint64_t *ls = reinterpret_cast<uint64_t *>(ls_n + 2);
    for (uint32_t b = threadIdx.x; b < 2u * n_parts; b += blockDim.x) lh[b] = 0;
    if (sa.list && threadIdx.x == 0) ls_n[0] = 0;
    __syncthreads();
    const SmerCfg cf = smer_cfg(k);
    const uint64_t nwords = (total + 15) / 16, nsteps = (nwords + SMER_STEP_WORDS - 1) / SMER_STEP_WORDS, nwm = nsteps * SMER_STEP_WORDS + 64;
    const uint64_t s0 = (uint64_t) blockIdx.x * steps_per_unit;
    const uint64_t s1 = s0 + steps_per_unit < nsteps ? s0 + steps_per_unit : nsteps;
    const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t bad = 0;
    SmerRaw raw;
    smer_fetch(bases, novalid, total, nwm, s0 + wave, raw);
    uint32_t accr[4] = {0, 0, 0, 0}, acck[4] = {0, 0, 0, 0}, since = 0;
    auto flush = [&]() { // the lanes' 16-bit fields -> LDS, one atomic per field and wave
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const uint32_t o = (uint32_t) (i >> 1) + 4u * (uint32_t) (i & 1) + 2u * (uint32_t) half;
                const uint32_t sr = wave_incl_scan_u32(half ? accr[i] >> 16 : accr[i] & 0xFFFFu);
                const uint32_t sk = wave_incl_scan_u32(half ? acck[i] >> 16 : acck[i] & 0xFFFFu);
                if (lane_id() == 63 && o < n_parts) {
                    if (sr) atomicAdd(&lh[o], sr);
                    if (sk) atomicAdd(&lk[o], sk);
                }
            }
            accr[i] = 0;
            acck[i] = 0;
        }
        since = 0;
    };
    for (uint64_t st = s0 + wave; st < s1; st += nwaves) {
        SmerLane a;
        uint32_t mw[16], own[16];
        const SmerRaw cur = raw;
        smer_fetch(bases, novalid, total, nwm, st + nwaves, raw); // the next step of this wave, in flight under this one's arithmetic
        smer_step<W>(bases, total, cf, n_parts, st, cur, a, mw, own);
        bad |= a.bad;
        uint64_t pr = 0, pk = 0;
        uint32_t rm = a.rm;
        OwnerOf<PACK8> oo;
        oo.set(own);
        while (rm) {
            const uint32_t s = (uint32_t) __builtin_ctz(rm);
            rm &= rm - 1u;
            const uint32_t L = smer_len(a, s), o = oo.at(own, s);
            if (PACK8) {
                pr += 1ull << (8u * o);
                pk += (uint64_t) L << (8u * o);
            } else {
                atomicAdd(&lh[o], 1u);
                atomicAdd(&lk[o], L);
            }
        }
        if (PACK8) {
            uint32_t w[4];
            spread8(pr, w);
#pragma unroll
            for (int i = 0; i < 4; i++) accr[i] += w[i];
            spread8(pk, w);
#pragma unroll
            for (int i = 0; i < 4; i++) acck[i] += w[i];
            if (++since == SMER_FLUSH_STEPS) flush();
        }
        if (sa.list) { // the duplication sample: the k-mers whose MINIMIZER is sampled (all occurrences of a k-mer or none)
            uint32_t sm = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) sm |= (smer_sampled(mw[j], sa.shift) ? 1u : 0u) << j;
            sm &= lane_id() < SMER_STEP_WORDS ? a.V : 0u;
            if (__any(sm != 0u)) {
                const uint64_t hi = ((uint64_t) a.w0 << 32) | a.w1;
                while (sm) {
                    const uint32_t j = (uint32_t) __builtin_ctz(sm);
                    sm &= sm - 1u;
                    const uint64_t v = ((hi << (2u * j)) | (((uint64_t) a.w2 << (2u * j)) >> 32)) >> (64 - 2 * k);
                    const uint64_t rc = revcomp_val(v, k);
                    const uint32_t at = atomicAdd(&ls_n[0], 1u);
                    if (at < SAMPLE_LDS) ls[at] = rc < v ? rc : v;
                }
            }
        }
    }
    if (PACK8) flush();
    if (bad) atomicOr(err, DERR_NON_ACGT);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < n_parts; b += blockDim.x) {
        hist[(uint64_t) blockIdx.x * n_parts + b] = lh[b];
        if (lk[b]) atomicAdd(&kmers_per_owner[b], (unsigned long long) lk[b]);
    }
    if (sa.list) {
        __shared__ uint32_t gbase;
        const uint32_t cnt = ls_n[0], keep = cnt < SAMPLE_LDS ? cnt : SAMPLE_LDS;
        if (threadIdx.x == 0) {
            gbase = atomicAdd(&sa.n[0], keep);
            if (cnt > SAMPLE_LDS) sa.n[1] = 1u; // the sample of this workgroup is truncated: the estimate is void
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < keep; i += blockDim.x)
            if (gbase + i < sa.cap) sa.list[gbase + i] = ls[i];
            else sa.n[1] = 1u;
    }
}

// offs[u][b] = records of units < u for owner b; tot[b] = records for owner b (one workgroup per owner)
__global__ void __launch_bounds__(256) k_smer_scan_a(const uint32_t *hist, uint32_t units, uint32_t n_parts, uint64_t *offs, uint64_t *tot) {
    __shared__ uint64_t part[256];
    const uint32_t b = blockIdx.x, per = (units + 255) / 256;
    const uint32_t u0 = threadIdx.x * per < units ? threadIdx.x * per : units, u1 = u0 + per < units ? u0 + per : units;
    uint64_t sum = 0;
    for (uint32_t u = u0; u < u1; u++) sum += hist[(uint64_t) u * n_parts + b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 256; i++) { const uint64_t v = part[i]; part[i] = run; run += v; }
        tot[b] = run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t u = u0; u < u1; u++) {
        offs[(uint64_t) u * n_parts + b] = run;
        run += hist[(uint64_t) u * n_parts + b];
    }
}
// binstart = exclusive scan of tot; binstart[n_parts] = all records (single thread: n_parts <= 2048)
__global__ void k_smer_scan_b(const uint64_t *tot, uint32_t n_parts, uint64_t *binstart) {
    if (threadIdx.x || blockIdx.x) return;
    uint64_t run = 0;
    for (uint32_t b = 0; b < n_parts; b++) { binstart[b] = run; run += tot[b]; }
    binstart[n_parts] = run;
}

template <int W, bool PACK8>
__global__ void __launch_bounds__(SMER_THREADS) k_smer_scatter(const uint8_t *bases, const uint16_t *novalid, uint64_t total, int k,
                                                                uint32_t n_parts, uint32_t steps_per_unit, const uint64_t *offs,
                                                                const uint64_t *binstart, uint32_t *out) {
    extern __shared__ __attribute__((aligned(8))) uint8_t smem[];
    uint64_t *ubase = reinterpret_cast<uint64_t *>(smem);       // [n_parts] first record of this unit's range of an owner
    uint32_t *lcur = reinterpret_cast<uint32_t *>(ubase + n_parts); // [n_parts] records of this unit handed out so far
    uint32_t *lwb = lcur + n_parts;                              // PACK8: [waves][8] a wave's base of a step
    for (uint32_t b = threadIdx.x; b < n_parts; b += blockDim.x) {
        ubase[b] = binstart[b] + offs[(uint64_t) blockIdx.x * n_parts + b];
        lcur[b] = 0;
    }
    __syncthreads();
    const SmerCfg cf = smer_cfg(k);
    const uint64_t nwords = (total + 15) / 16, nsteps = (nwords + SMER_STEP_WORDS - 1) / SMER_STEP_WORDS, nwm = nsteps * SMER_STEP_WORDS + 64;
    const uint64_t s0 = (uint64_t) blockIdx.x * steps_per_unit;
    const uint64_t s1 = s0 + steps_per_unit < nsteps ? s0 + steps_per_unit : nsteps;
    const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, lane = (uint32_t) lane_id();
    SmerRaw raw;
    smer_fetch(bases, novalid, total, nwm, s0 + wave, raw);
    for (uint64_t st = s0 + wave; st < s1; st += nwaves) {
        SmerLane a;
        uint32_t mw[16], own[16];
        const SmerRaw cur = raw;
        smer_fetch(bases, novalid, total, nwm, st + nwaves, raw);
        smer_step<W>(bases, total, cf, n_parts, st, cur, a, mw, own);
        uint32_t ex[4] = {0, 0, 0, 0}; // PACK8: records of the lower lanes of this step, per owner
        OwnerOf<PACK8> oo;
        oo.set(own);
        if (PACK8) {
            uint64_t pr = 0;
            uint32_t rm = a.rm;
            while (rm) {
                const uint32_t s = (uint32_t) __builtin_ctz(rm);
                rm &= rm - 1u;
                pr += 1ull << (8u * oo.at(own, s));
            }
            uint32_t w[4], tot[4];
            spread8(pr, w);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t inc = wave_incl_scan_u32(w[i]);
                ex[i] = inc - w[i];
                tot[i] = (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
            }
            if (lane < n_parts) { // lane o takes owner o's range of the step from the unit's cursor
                const uint32_t t = field16(tot, lane);
                lwb[wave * 8u + lane] = t ? atomicAdd(&lcur[lane], t) : 0u;
            }
        }
        uint64_t seen = 0;
        uint32_t rm = a.rm;
        const uint64_t A = ((uint64_t) a.w0 << 32) | a.w1, Bw = ((uint64_t) a.w2 << 32) | a.w3;
        while (rm) {
            const uint32_t s = (uint32_t) __builtin_ctz(rm);
            rm &= rm - 1u;
            const uint32_t L = smer_len(a, s), o = oo.at(own, s);
            uint64_t pos;
            if (PACK8) {
                pos = ubase[o] + lwb[wave * 8u + o] + field16(ex, o) + (uint32_t) ((seen >> (8u * o)) & 0xFFu);
                seen += 1ull << (8u * o);
            } else pos = ubase[o] + atomicAdd(&lcur[o], 1u);
            // the record: bases s .. s + L + k - 2 of the lane's 64-base window, left-aligned in 96 bits, zeros behind
            uint64_t h64 = s ? (A << (2u * s)) | (Bw >> (64u - 2u * s)) : A;
            uint32_t r2 = (uint32_t) ((Bw << (2u * s)) >> 32);
            const uint32_t nbits = 2u * (L + (uint32_t) cf.k - 1u); // <= 92
            if (nbits >= 64u) r2 = nbits > 64u ? r2 & (0xFFFFFFFFu << (96u - nbits)) : 0u;
            else {
                h64 &= ~0ull << (64u - nbits);
                r2 = 0u;
            }
            uint32_t *dst = out + pos * 3u;
            dst[0] = (uint32_t) (h64 >> 32);
            dst[1] = (uint32_t) h64;
            dst[2] = r2 | (L - 1u);
        }
    }
}

// ---- receiver-side helpers that need no table -------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_smer_count_kmers(const uint32_t *recs, uint64_t n, unsigned long long *out) {
    uint64_t mine = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) mine += (recs[i * 3 + 2] & 15u) + 1u;
    const uint32_t s = wave_incl_scan_u32((uint32_t) mine); // (a thread sees < 2^26 k-mers per launch: grids of >= 2 048 threads)
    if (lane_id() == 63 && s) atomicAdd(out, (unsigned long long) s);
}
// records -> canonical k-mers (small batches and the fall-back of the receiver: the order does not matter to a counter)
__global__ void __launch_bounds__(256) k_smer_expand(const uint32_t *recs, uint64_t n, int k, uint64_t *out, unsigned long long *cursor) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x, rounds = (n + stride - 1) / stride; // (wave-uniform: scans inside)
    for (uint64_t it = 0; it < rounds; it++) {
        const uint64_t i = it * stride + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
        SmerRec r = {{0u, 0u, 0u}};
        uint32_t L = 0;
        if (i < n) {
            r.w[0] = recs[i * 3]; r.w[1] = recs[i * 3 + 1]; r.w[2] = recs[i * 3 + 2];
            L = smer_rec_kmers(r);
        }
        const uint32_t inc = wave_incl_scan_u32(L);
        const uint32_t tot = (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
        unsigned long long base = 0;
        if (lane_id() == 0 && tot) base = atomicAdd(cursor, (unsigned long long) tot);
        base = ((unsigned long long) bcast_u32((uint32_t) (base >> 32), 0) << 32) | bcast_u32((uint32_t) base, 0);
        uint64_t *dst = out + base + (inc - L);
        for (uint32_t j = 0; j < L; j++) {
            const uint64_t v = smer_rec_kmer(r, k, j), rc = revcomp_val(v, k);
            dst[j] = rc < v ? rc : v;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
uint32_t smer_units(const kmu_ctx *ctx, uint64_t total_bases) {
    const uint64_t nwords = (total_bases + 15) / 16, nsteps = std::max<uint64_t>(1, (nwords + SMER_STEP_WORDS - 1) / SMER_STEP_WORDS);
    return (uint32_t) std::min<uint64_t>(nsteps, (uint64_t) ctx->num_cus * 8);
}

template <typename F>
static int smer_dispatch(int w, bool pack8, F &&f) { // f(W, PACK8 as integral constants)
#define KMU_SMER_CASE(WV)                                                                  \
    if (w == WV) return pack8 ? f(std::integral_constant<int, WV>(), std::true_type()) : f(std::integral_constant<int, WV>(), std::false_type());
    KMU_SMER_CASE(21)
    KMU_SMER_CASE(16)
    KMU_SMER_CASE(9)
#undef KMU_SMER_CASE
    return KMU_E_BAD_ARG;
}

int smer_census(kmu_ctx *ctx, const DevSeqs &ds, uint64_t total_bases, int k, uint32_t n_parts, uint32_t *d_err, const SampleArgs &sa,
                SmerGroups *g) {
    if (n_parts == 0 || n_parts > 2048) return fail(ctx, KMU_E_BAD_ARG, "n_parts must be in 1..2048");
    if (!smer_supported(KMU_KMER64BIT, k)) return fail(ctx, KMU_E_UNSUPPORTED, "minimizer owners need Kmer64bit with 17 <= k <= 31");
    if (total_bases == 0 || ds.n_seq == 0) { // an empty shard: no record for anybody
        g->k = k;
        g->n_parts = n_parts;
        g->units = 0;
        g->steps_per_unit = 0;
        KMU_TRY(dev_buf(ctx, "smer.binstart", ((size_t) n_parts + 1) * 8 + 64, &g->binstart));
        KMU_TRY(dev_buf(ctx, "smer.kmers", (size_t) n_parts * 8 + 64, &g->kmers));
        KMU_HIP(ctx, hipMemsetAsync(g->binstart, 0, ((size_t) n_parts + 1) * 8, ctx->stream));
        KMU_HIP(ctx, hipMemsetAsync(g->kmers, 0, (size_t) n_parts * 8, ctx->stream));
        return KMU_OK;
    }
    const uint64_t nwords = (total_bases + 15) / 16, nsteps = std::max<uint64_t>(1, (nwords + SMER_STEP_WORDS - 1) / SMER_STEP_WORDS);
    uint32_t units = smer_units(ctx, total_bases);
    const uint32_t spu = (uint32_t) ((nsteps + units - 1) / units);
    units = (uint32_t) ((nsteps + spu - 1) / spu);
    g->k = k;
    g->n_parts = n_parts;
    g->units = units;
    g->steps_per_unit = spu;
    void *tot;
    KMU_TRY(dev_buf(ctx, "smer.hist", (size_t) units * n_parts * 4 + 64, &g->hist));
    KMU_TRY(dev_buf(ctx, "smer.offs", (size_t) units * n_parts * 8 + 64, &g->offs));
    KMU_TRY(dev_buf(ctx, "smer.tot", (size_t) n_parts * 8 + 64, &tot));
    KMU_TRY(dev_buf(ctx, "smer.binstart", ((size_t) n_parts + 1) * 8 + 64, &g->binstart));
    KMU_TRY(dev_buf(ctx, "smer.kmers", (size_t) n_parts * 8 + 64, &g->kmers));
    KMU_HIP(ctx, hipMemsetAsync(g->kmers, 0, (size_t) n_parts * 8, ctx->stream));
    // the positions that start no k-mer: 16 bits per code word of every wave step incl. the last one's halo
    KMU_TRY(flat_novalid(ctx, ds, total_bases, k, nsteps * SMER_STEP_WORDS + 64, "smer.novalid", &g->novalid));
    g->total = total_bases;
    const size_t lds = ((size_t) 2 * n_parts + 4) * 4 + (sa.list ? (size_t) SAMPLE_LDS * 8 : 0);
    const SmerCfg cf = smer_cfg(k);
    {
        KernelTimer tm(ctx, "k_smer_census");
        KMU_TRY(smer_dispatch(cf.w, n_parts <= 8, [&](auto W, auto P8) {
            hipLaunchKernelGGL((k_smer_census<decltype(W)::value, decltype(P8)::value>), dim3(units), dim3(SMER_THREADS), lds, ctx->stream, ds.bases,
                               (const uint16_t *) g->novalid, total_bases, k, n_parts, spu, (uint32_t *) g->hist, (unsigned long long *) g->kmers, d_err, sa);
            return (int) KMU_OK;
        }));
    }
    {
        KernelTimer tm(ctx, "k_smer_scan");
        hipLaunchKernelGGL(k_smer_scan_a, dim3(n_parts), dim3(256), 0, ctx->stream, (const uint32_t *) g->hist, units, n_parts, (uint64_t *) g->offs,
                           (uint64_t *) tot);
        hipLaunchKernelGGL(k_smer_scan_b, dim3(1), dim3(64), 0, ctx->stream, (const uint64_t *) tot, n_parts, (uint64_t *) g->binstart);
    }
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

int flat_novalid(kmu_ctx *ctx, const DevSeqs &ds, uint64_t total_bases, int k, uint64_t n_words, const char *buf_name, void **out) {
    const uint64_t mask_bytes = (n_words * 2 + 3) & ~(uint64_t) 3;
    KMU_TRY(dev_buf(ctx, buf_name, (size_t) mask_bytes + 64, out));
    KMU_HIP(ctx, hipMemsetAsync(*out, 0, (size_t) mask_bytes, ctx->stream));
    hipLaunchKernelGGL(k_smer_novalid, dim3((unsigned) (((uint64_t) ds.n_seq + 2 + 255) / 256)), dim3(256), 0, ctx->stream, ds.offsets, ds.n_seq, k,
                       total_bases, (uint64_t) mask_bytes * 8, (uint32_t *) *out);
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

int smer_scatter(kmu_ctx *ctx, const DevSeqs &ds, uint64_t total_bases, const SmerGroups &g, void *records_out) {
    if (g.units == 0 || total_bases != g.total) return g.units == 0 ? (int) KMU_OK : fail(ctx, KMU_E_BAD_ARG, "smer_scatter: not the batch of the census");
    const size_t lds = (size_t) g.n_parts * 12 + 4 * 8 * 4 + 16;
    const SmerCfg cf = smer_cfg(g.k);
    KernelTimer tm(ctx, "k_smer_scatter");
    KMU_TRY(smer_dispatch(cf.w, g.n_parts <= 8, [&](auto W, auto P8) {
        hipLaunchKernelGGL((k_smer_scatter<decltype(W)::value, decltype(P8)::value>), dim3(g.units), dim3(SMER_THREADS), lds, ctx->stream, ds.bases,
                           (const uint16_t *) g.novalid, g.total, g.k, g.n_parts, g.steps_per_unit, (const uint64_t *) g.offs,
                           (const uint64_t *) g.binstart, (uint32_t *) records_out);
        return (int) KMU_OK;
    }));
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

int smer_count_kmers(kmu_ctx *ctx, const void *records, uint64_t n, uint64_t *d_out) {
    KMU_HIP(ctx, hipMemsetAsync(d_out, 0, 8, ctx->stream));
    if (!n) return KMU_OK;
    const int grid = (int) std::min<uint64_t>(std::max<uint64_t>((n + 255) / 256, 8), (uint64_t) ctx->num_cus * 8);
    hipLaunchKernelGGL(k_smer_count_kmers, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t *) records, n, (unsigned long long *) d_out);
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

int smer_expand(kmu_ctx *ctx, const void *records, uint64_t n, int k, uint64_t *out, uint64_t *d_cursor) {
    KMU_HIP(ctx, hipMemsetAsync(d_cursor, 0, 8, ctx->stream));
    if (!n) return KMU_OK;
    const int grid = (int) std::min<uint64_t>((n + 255) / 256, (uint64_t) ctx->num_cus * 8);
    KernelTimer tm(ctx, "k_smer_expand");
    hipLaunchKernelGGL(k_smer_expand, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t *) records, n, k, out, (unsigned long long *) d_cursor);
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

} // namespace kmu

extern "C" int kmu_kmer_owner_minimizer(int kmer_size, const uint64_t *canon_kmers, uint64_t n, uint32_t n_parts, uint32_t *owners_out) {
    if ((!canon_kmers || !owners_out) && n) return KMU_E_BAD_ARG;
    if (n_parts == 0 || !kmu::smer_supported(KMU_KMER64BIT, kmer_size)) return KMU_E_BAD_ARG;
    for (uint64_t i = 0; i < n; i++) owners_out[i] = kmu::smer_owner_of_kmer(canon_kmers[i], kmer_size, n_parts);
    return KMU_OK;
}
