// micro-benchmark: what the L2 / fabric make of the write pattern of the single-pass partition -- 256 workgroups append short runs of
// 8-byte items to shared streams [set][bin] through one atomic per bin and tile on the stream's cursor (kmu_count_part.hip,
// tile_scatter_seg's write-out).  Run it under `rocprofv3 --pmc WRITE_SIZE` to see the bytes the fabric counts per mode.
//   usage: append_runs <mean run, items> <mode> [sets=16] [bins=2048] [item bytes = 8 | 6]
//   mode 0: runs of R/2 .. 3R/2 items wherever the cursor stands (the product's pattern)
//   mode 1: runs of exactly R items, R a multiple of 8: every run starts on a 64-byte line (what a carry of < 8 items per bin would give)
//   mode 2: runs of a multiple of 4 items (32-byte sectors)
//   mode 3: as 0, every workgroup its own streams (sets = workgroups)
//   mode 4: as 0 with non-temporal stores
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static constexpr int THREADS = 1024;
__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
template <int MODE, int BYTES>
__global__ void __launch_bounds__(THREADS) k_append(uint8_t *out, uint32_t *cursor, uint32_t bins, uint32_t sets, uint32_t R, uint32_t tiles, uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *lstart = reinterpret_cast<uint32_t *>(smem);        // bins + 1
    uint32_t *grel = lstart + bins + 1;                           // bins
    uint16_t *binof = reinterpret_cast<uint16_t *>(grel + bins);  // positions of the tile
    __shared__ uint32_t wtot[16];
    const uint32_t tid = threadIdx.x, set = MODE == 3 ? blockIdx.x : blockIdx.x % sets;
    uint32_t *cur = cursor + (size_t) set * bins;
    for (uint32_t t = 0; t < tiles; t++) {
        // two bins per thread: lengths, cursor, scan
        uint32_t len[2] = {0, 0}, run[2] = {0, 0};
        for (int e = 0; e < 2; e++) {
            const uint32_t b = 2 * tid + e;
            if (b < bins) {
                const uint32_t h = mix(b * 2654435761u + t * 40503u + blockIdx.x * 9176u);
                if (MODE == 1) len[e] = R;
                else if (MODE == 2) len[e] = ((R / 2 + h % (R + 1)) + 2) & ~3u;
                else len[e] = R / 2 + h % (R + 1);
                run[e] = atomicAdd(&cur[b], len[e]);
            }
        }
        uint32_t s = len[0] + len[1], incl = s;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if ((tid & 63) >= (uint32_t) d) incl += o; }
        if ((tid & 63) == 63) wtot[tid >> 6] = incl;
        __syncthreads();
        uint32_t wpre = 0, total = 0;
        for (int w = 0; w < 16; w++) { if ((uint32_t) w < (tid >> 6)) wpre += wtot[w]; total += wtot[w]; }
        const uint32_t excl = wpre + incl - s;
        if (2 * tid < bins) {
            lstart[2 * tid] = excl; grel[2 * tid] = run[0] - excl;
            for (uint32_t p = 0; p < len[0]; p++) binof[excl + p] = (uint16_t) (2 * tid);
        }
        if (2 * tid + 1 < bins) {
            lstart[2 * tid + 1] = excl + len[0]; grel[2 * tid + 1] = run[1] - (excl + len[0]);
            for (uint32_t p = 0; p < len[1]; p++) binof[excl + len[0] + p] = (uint16_t) (2 * tid + 1);
        }
        __syncthreads();
        for (uint32_t p = tid; p < total; p += THREADS) {
            const uint32_t b = binof[p];
            const uint32_t rel = grel[b] + p;
            if (rel < cap) {
                const uint64_t at = (uint64_t) (set * bins + b) * cap + rel;
                const uint64_t v = ((uint64_t) t << 32) | p;
                if (BYTES == 8) {
                    uint64_t *o = reinterpret_cast<uint64_t *>(out) + at;
                    if (MODE == 4) __builtin_nontemporal_store(v, o);
                    else *o = v;
                } else {
                    uint8_t *blk = out + (at >> 3) * 48u;
                    reinterpret_cast<uint32_t *>(blk)[at & 7u] = (uint32_t) v;
                    reinterpret_cast<uint16_t *>(blk + 32)[at & 7u] = (uint16_t) (v >> 32);
                }
            }
        }
        __syncthreads();
    }
}

int main(int argc, char **argv) {
    const uint32_t R = argc > 1 ? atoi(argv[1]) : 8;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    uint32_t sets = argc > 3 ? atoi(argv[3]) : 16;
    const uint32_t bins = argc > 4 ? atoi(argv[4]) : 2048;
    const int bytes = argc > 5 ? atoi(argv[5]) : 8;
    const uint32_t wgs = 256;
    if (mode == 3) sets = wgs;
    const uint64_t n_target = 1ull << 31; // items per launch (16 GiB of 8-byte items)
    const uint32_t tile = bins * R;
    if (tile > 24576 || bins > 2048) { printf("tile too big\n"); return 1; }
    const uint32_t tiles = (uint32_t) (n_target / wgs / tile);
    const uint64_t n = (uint64_t) tiles * wgs * tile; // expected items (mode 0: on average)
    const uint64_t per_stream = n / sets / bins;
    const uint32_t cap = (uint32_t) ((per_stream + per_stream / 8 + 4096 + 15) & ~15ull);
    const size_t out_bytes = (size_t) sets * bins * cap * 8;
    uint8_t *out; uint32_t *cursor;
    CK(hipMalloc(&out, out_bytes));
    CK(hipMalloc(&cursor, (size_t) sets * bins * 4));
    const size_t lds = (size_t) (2 * bins + 1) * 4 + (size_t) (tile * 3 / 2 + 64) * 2 + 65536; // (+ 64 KiB: one workgroup per CU)
    auto kern = bytes == 6 ? (mode == 1 ? k_append<1, 6> : mode == 2 ? k_append<2, 6> : mode == 3 ? k_append<3, 6> : k_append<0, 6>)
                           : (mode == 1 ? k_append<1, 8> : mode == 2 ? k_append<2, 8> : mode == 3 ? k_append<3, 8> : mode == 4 ? k_append<4, 8> : k_append<0, 8>);
    CK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 3; it++) {
        CK(hipMemset(cursor, 0, (size_t) sets * bins * 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(THREADS), lds, 0, out, cursor, bins, sets, R, tiles, cap);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<uint32_t> h((size_t) sets * bins);
    CK(hipMemcpy(h.data(), cursor, h.size() * 4, hipMemcpyDeviceToHost));
    uint64_t items = 0, over = 0;
    for (uint32_t c : h) { items += c < cap ? c : cap; over += c > cap; }
    printf("R %u mode %d sets %u bins %u bytes %d: items %.3e = %.2f GB in %.2f ms = %.2f TB/s (streams over capacity: %llu)\n", R, mode, sets, bins, bytes,
           (double) items, items * (double) bytes / 1e9, best, items * (double) bytes / 1e9 / best, (unsigned long long) over);
    return 0;
}
