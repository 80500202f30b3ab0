// kmu_flat.h -- the reads of a batch as ONE flat stream of bases: which read holds a given base (device code).
#pragma once

#include "kmu_device.h"

namespace kmu {

// largest i with offsets[i] <= g, g wave-uniform; 64-ary search, one coalesced probe per round
__device__ __forceinline__ uint32_t wave_find_read(const uint64_t *offsets, uint32_t n, uint64_t g) {
    uint32_t lo = 0, hi = n; // invariant offsets[lo] <= g < offsets[hi]
    const uint32_t lane = (uint32_t) lane_id();
    while (hi - lo > 1) {
        const uint32_t step = (hi - lo + 63) / 64;
        const uint64_t idx = (uint64_t) lo + (uint64_t) (lane + 1) * step;
        const bool le = idx < hi && offsets[idx] <= g;
        const uint32_t c = (uint32_t) __popcll(__ballot(le));
        const uint64_t nhi = (uint64_t) lo + (uint64_t) (c + 1) * step;
        lo = lo + c * step;
        hi = nhi < hi ? (uint32_t) nhi : hi;
    }
    return lo;
}

// the same with a hint: a wave walks the flat stream forwards, so the read is usually one of the next 64
__device__ __forceinline__ uint32_t wave_find_read_from(const uint64_t *offsets, uint32_t n, uint64_t g, uint32_t hint) {
    if (hint >= n || offsets[hint] > g) return wave_find_read(offsets, n, g);
    const uint64_t idx = (uint64_t) hint + 1 + (uint32_t) lane_id();
    const bool le = idx < n && offsets[idx] <= g;
    const uint32_t c = (uint32_t) __popcll(__ballot(le));
    if (c < 64u) return hint + c;
    return wave_find_read(offsets, n, g);
}

// One wave step (64 words = 1024 bases) of the flat base stream: f(canon) for every k-mer that lies inside one read
// (kmer.reverse_complement().min(kmer), kmercount.rs:938).  Returns a non-zero mask if this lane saw a non-ACGT byte.
// The reads occupy [offsets[0], total) of the stream: `start` = offsets[0] need not be 0 (a range of a larger read set).
template <typename F>
__device__ __forceinline__ uint32_t flat_step_canon(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq,
                                                    uint64_t total, uint64_t start, int k, uint64_t st, uint32_t &r_hint, F &&f) {
    SeqView s;
    s.base = bases; s.begin = 0; s.len = total; s.total = total; s.packed = 0;
    const int lane = lane_id();
    const uint64_t widx = st * 64 + lane;
    uint32_t bad, bad2;
    uint32_t w0 = load_code_word(s, widx, bad);
    uint32_t ex = load_code_word(s, st * 64 + 64 + (uint64_t) (lane & 1), bad2);
    uint32_t e0 = bcast_u32(ex, 0), e1 = bcast_u32(ex, 1);
    uint32_t w1 = shfl_down_u32(w0, 1), w2 = shfl_down_u32(w0, 2);
    if (lane == 63) { w1 = e0; w2 = e1; }
    if (lane == 62) { w2 = e0; }
    uint32_t r = wave_find_read_from(offsets, n_seq, st * 1024 < total ? st * 1024 : total - 1, r_hint);
    r_hint = r;
    const uint64_t g0 = widx * 16;
    const bool in = g0 < total && g0 + 16 > start;
    uint64_t rend = 0;
    if (in) {
        rend = offsets[r + 1];
        while (g0 >= rend && r + 1 < n_seq) { r++; rend = offsets[r + 1]; } // the read of this lane's first base
    }
    const uint64_t hi = ((uint64_t) w0 << 32) | w1;
    const int sh = 64 - 2 * k;
    // A wave whose lanes all sit well inside a read (long reads: most waves) skips the per-k-mer boundary tests.
    if (__all(!in || (g0 >= start && rend - g0 >= (uint64_t) (15 + k)))) {
        if (in) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint64_t val = ((hi << (2 * j)) | (((uint64_t) w2 << (2 * j)) >> 32)) >> sh, rc = revcomp_val(val, k);
                f(rc < val ? rc : val);
            }
        }
    } else if (in) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint64_t g = g0 + j;
            while (g >= rend && r + 1 < n_seq) { r++; rend = offsets[r + 1]; }
            if (g >= start && g + k <= rend) {
                const uint64_t val = ((hi << (2 * j)) | (((uint64_t) w2 << (2 * j)) >> 32)) >> sh, rc = revcomp_val(val, k);
                f(rc < val ? rc : val);
            }
        }
    }
    return bad;
}

} // namespace kmu
