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

} // namespace kmu
