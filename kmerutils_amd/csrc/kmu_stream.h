// kmu_stream.h -- wave-level streaming of the k-mers of one sequence (device code).
//
// KmerSeqIterator::next (src/base/kmergenerator.rs:75-106) yields the L-k+1 forward k-mers of a sequence one
// base at a time.  Here one wavefront covers 64 x 16 = 1024 consecutive k-mer start positions per step: lane l
// owns the 16 positions that start inside its aligned 16-base code word and extracts each k-mer (and its reverse
// complement) directly from a 48-base window (its own word + the next two, fetched from the neighbouring lanes
// by wave shuffles).  No loop-carried roll, no LDS staging, no barrier.
#pragma once

#include "kmu_device.h"

namespace kmu {

// number of aligned code words that hold the sequence
__device__ __forceinline__ uint64_t seq_num_words(const SeqView &s) { return (seq_lead(s) + s.len + 15) / 16; }

// Visit the k-mers with start position in [pos_begin, pos_end) (pos_end <= L-k+1) of the words of wave-step
// `step` (64 words).  f(pos, val, rcval) is called for every valid position of this lane.
// Returns a non-zero mask if this lane saw a non-ACGT byte inside the sequence.
template <typename F>
__device__ __forceinline__ uint32_t wave_step_kmers(const SeqView &s, int k, uint64_t step, uint64_t pos_begin,
                                                    uint64_t pos_end, F &&f) {
    const int lane = lane_id();
    const uint64_t widx = step * 64 + (uint64_t) lane;
    uint32_t bad, bad_halo;
    uint32_t w0 = load_code_word(s, widx, bad);
    uint32_t ex = load_code_word(s, step * 64 + 64 + (uint64_t) (lane & 1), bad_halo);
    uint32_t e0 = bcast_u32(ex, 0), e1 = bcast_u32(ex, 1);
    uint32_t w1 = shfl_down_u32(w0, 1);
    uint32_t w2 = shfl_down_u32(w0, 2);
    if (lane == 63) { w1 = e0; w2 = e1; }
    if (lane == 62) { w2 = e0; }
    const uint64_t hi = ((uint64_t) w0 << 32) | w1;
    const uint32_t lead = seq_lead(s);
    const int64_t p0 = (int64_t) (widx * 16) - (int64_t) lead;
    const int sh = 64 - 2 * k;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        int64_t p = p0 + j;
        if (p >= (int64_t) pos_begin && p < (int64_t) pos_end) {
            uint64_t v = (hi << (2 * j)) | (((uint64_t) w2 << (2 * j)) >> 32);
            uint64_t val = v >> sh;
            uint64_t rc = revcomp_val(val, k);
            f((uint64_t) p, val, rc);
        }
    }
    return bad | bad_halo; // halo words are validated too: every base used by a k-mer is checked
}

// walk all words of a sequence only to validate its bytes (sequences shorter than k yield no k-mer)
__device__ __forceinline__ uint32_t wave_validate_seq(const SeqView &s, int wave, int nwaves, bool aa) {
    uint32_t bad = 0;
    if (aa) {
        for (uint64_t p = (uint64_t) wave * 64 + lane_id(); p < s.len; p += (uint64_t) nwaves * 64)
            bad |= code_aa(s.base[s.begin + p]) == 0;
        return bad;
    }
    uint64_t nw = seq_num_words(s);
    for (uint64_t w = (uint64_t) wave * 64 + lane_id(); w < nw; w += (uint64_t) nwaves * 64) {
        uint32_t b;
        (void) load_code_word(s, w, b);
        bad |= b;
    }
    return bad;
}

// amino-acid sequences are stored one byte per residue (src/aautils/kmeraa.rs:404-484); k <= 12.
// One position per lane per step of 64.  f(pos, val, 0).  Returns non-zero on an invalid residue.
template <typename F>
__device__ __forceinline__ uint32_t wave_step_kmers_aa(const SeqView &s, int k, uint64_t step, uint64_t pos_begin,
                                                       uint64_t pos_end, F &&f) {
    uint64_t p = step * 64 + (uint64_t) lane_id();
    uint32_t bad = 0;
    if (p < s.len) bad = code_aa(s.base[s.begin + p]) == 0;
    if (p >= pos_begin && p < pos_end) {
        uint64_t val = 0;
        for (int j = 0; j < k; j++) {
            uint32_t c = code_aa(s.base[s.begin + p + j]);
            val = (val << 5) | c;
        }
        f(p, val, 0ull);
    }
    return bad;
}

} // namespace kmu
