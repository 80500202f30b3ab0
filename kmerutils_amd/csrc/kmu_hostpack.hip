// kmu_hostpack.hip -- bases cross PCIe at 2 bits: the host side of kmu_sketch_count's pipeline on host buffers.
//
// Reference: the reader thread of datasketcher packs every read with Sequence::new(raw, 2) before anything else happens
// (src/bin/datasketcher.rs:358-388, src/base/sequence.rs:25-106) -- serially, one read at a time.  Here the host's cores pack the
// caller's ASCII bases into the same 4-bases-per-byte form (first base in bits 7..6) with AVX2, chunk by chunk ahead of the
// upload, a quarter of the bytes crosses PCIe (4.38 GB at ~55 GB/s was 80 ms of the headline's host-to-host step), and a small
// kernel on the upload stream turns a chunk back into the ASCII stream every kernel of the path consumes.  The packer is also
// where a byte outside ACGTacgt is found (the reference panics in Alphabet2b::encode, src/base/alphabet.rs:125): what comes out
// of the unpack kernel is ACGT by construction.
//
// Measured on the GPU box's host (scripts/micro/host_pack.cpp): 150 GB/s of bases with 8 threads, 220 with 16 -- three to four
// times what PCIe delivers, so the pipeline behind it is bound by the GPU's kernels.
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "kmu_hostpack.hpp"

namespace kmu {

static inline uint32_t code_of(uint8_t c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}
// n bases (any n) -> ceil(n / 4) bytes, zero padding in the last; returns false if a byte is outside ACGTacgt
static bool pack_scalar(const uint8_t *in, uint64_t n, uint8_t *out) {
    uint32_t bad = 0;
    uint64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const uint32_t a = code_of(in[i]), b = code_of(in[i + 1]), c = code_of(in[i + 2]), d = code_of(in[i + 3]);
        bad |= (a | b | c | d) & 4u;
        out[i >> 2] = (uint8_t) ((a << 6) | ((b & 3u) << 4) | ((c & 3u) << 2) | (d & 3u));
    }
    if (i < n) {
        uint32_t w = 0;
        for (uint64_t j = i; j < n; j++) {
            const uint32_t a = code_of(in[j]);
            bad |= a & 4u;
            w |= (a & 3u) << (6 - 2 * (j - i));
        }
        out[i >> 2] = (uint8_t) w;
    }
    return bad == 0;
}
// 32 bases -> 8 bytes
__attribute__((target("avx2"))) static inline bool pack32_avx2(const uint8_t *in, uint8_t *out) {
    const __m256i v = _mm256_loadu_si256((const __m256i *) in);
    const __m256i u = _mm256_and_si256(v, _mm256_set1_epi8((char) 0xDF));
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(u, _mm256_set1_epi8('C'))),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(u, _mm256_set1_epi8('T'))));
    // code = x ^ (x >> 1) with x = (c >> 1) & 3: A 0, C 1, G 2, T 3 (alphabet.rs:119-127, either case)
    const __m256i x = _mm256_and_si256(_mm256_srli_epi16(v, 1), _mm256_set1_epi8(3));
    const __m256i c = _mm256_xor_si256(x, _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(1)));
    const __m256i p = _mm256_maddubs_epi16(c, _mm256_set1_epi16(0x0104));   // b0 * 4 + b1 in every 16-bit lane
    const __m256i q = _mm256_madd_epi16(p, _mm256_set1_epi32(0x00010010)); // (pair 0) * 16 + pair 1 in every 32-bit lane: one byte
    const __m256i s = _mm256_shuffle_epi8(q, _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                                               0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
    const uint32_t lo = (uint32_t) _mm256_extract_epi32(s, 0), hi = (uint32_t) _mm256_extract_epi32(s, 4);
    memcpy(out, &lo, 4);
    memcpy(out + 4, &hi, 4);
    return _mm256_movemask_epi8(ok) == -1;
}
__attribute__((target("avx2"))) static bool pack_avx2(const uint8_t *in, uint64_t n, uint8_t *out) {
    bool ok = true;
    uint64_t i = 0;
    for (; i + 32 <= n; i += 32) ok &= pack32_avx2(in + i, out + (i >> 2));
    if (i < n) ok &= pack_scalar(in + i, n - i, out + (i >> 2));
    return ok;
}
bool host_pack2b(const uint8_t *in, uint64_t n, uint8_t *out) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return avx2 ? pack_avx2(in, n, out) : pack_scalar(in, n, out);
}

// ---- the packer of a pipelined call: worker threads take slabs of the stream in order; what has been packed is a growing prefix ----
struct PackPipe::Impl {
    const uint8_t *in = nullptr;
    uint8_t *out = nullptr;
    uint64_t total = 0;
    std::vector<uint64_t> slab_begin, slab_end;
    std::vector<uint8_t> done;
    size_t prefix = 0; // slabs [0, prefix) are packed
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::thread> workers;
    void run() {
        for (;;) {
            const size_t s = next.fetch_add(1);
            if (s >= slab_begin.size()) return;
            if (!host_pack2b(in + slab_begin[s], slab_end[s] - slab_begin[s], out + (slab_begin[s] >> 2))) bad.store(1);
            std::lock_guard<std::mutex> g(mu);
            done[s] = 1;
            const size_t before = prefix;
            while (prefix < done.size() && done[prefix]) prefix++;
            if (prefix != before) cv.notify_all();
        }
    }
};

PackPipe::PackPipe(const uint8_t *bases, uint8_t *packed_out, uint64_t total, int threads) {
    im = new Impl();
    im->in = bases;
    im->out = packed_out;
    im->total = total;
    const uint64_t slab = 4ull << 20; // (a multiple of 32 bases: slabs start on whole packed bytes and whole AVX2 steps)
    for (uint64_t b = 0; b < total; b += slab) {
        im->slab_begin.push_back(b);
        im->slab_end.push_back(std::min(total, b + slab));
    }
    im->done.assign(im->slab_begin.size(), 0);
    const int T = std::max(1, std::min<int>(threads, (int) std::max<size_t>(1, im->slab_begin.size())));
    for (int t = 0; t < T; t++) im->workers.emplace_back([this] { im->run(); });
}
bool PackPipe::wait_prefix(uint64_t end) {
    end = std::min(end, im->total);
    std::unique_lock<std::mutex> g(im->mu);
    im->cv.wait(g, [&] { return im->prefix == im->done.size() || im->slab_begin[im->prefix] >= end; });
    return im->bad.load() == 0;
}
PackPipe::~PackPipe() {
    im->next.store(im->slab_begin.size()); // (a call that ends early: the workers stop at their next slab)
    for (auto &t : im->workers) t.join();
    delete im;
}

// ---- device: 4 packed bytes -> 16 ASCII bases ----
__device__ __forceinline__ uint4 unpack16(uint32_t w) { // bytes in memory order: byte b holds bases 4 b .. 4 b + 3, the first in bits 7..6
    uint32_t o[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const uint32_t by = (w >> (8 * b)) & 0xFFu;
        uint32_t r = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) r |= ((0x54474341u >> (8u * ((by >> (6 - 2 * j)) & 3u))) & 0xFFu) << (8 * j); // "ACGT"[code]
        o[b] = r;
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// four words per thread and turn, their loads in flight together (one word per turn: 2.0 TB/s for the 5.5 GB of a 4.38 G base stream)
__global__ void __launch_bounds__(256) k_unpack2b(const uint32_t *packed, uint64_t n_bases, uint8_t *out) {
    constexpr int U = 4;
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x, nw = (n_bases + 15) / 16, whole = n_bases / 16;
    for (uint64_t i0 = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i0 < nw; i0 += stride * U) {
        uint32_t w[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = i0 + (uint64_t) u * stride;
            w[u] = i < nw ? packed[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = i0 + (uint64_t) u * stride;
            if (i >= nw) break;
            const uint4 o4 = unpack16(w[u]);
            if (i < whole) *reinterpret_cast<uint4 *>(out + i * 16) = o4;
            else {
                const uint32_t o[4] = {o4.x, o4.y, o4.z, o4.w};
                for (uint64_t t = i * 16; t < n_bases; t++) out[t] = (uint8_t) (o[(t >> 2) & 3] >> (8 * (t & 3)));
            }
        }
    }
}

int launch_unpack2b(kmu_ctx *ctx, const void *packed_dev, uint64_t n_bases, uint8_t *out_dev, hipStream_t s) {
    if (!n_bases) return KMU_OK;
    const uint64_t nw = (n_bases + 15) / 16;
    const int grid = (int) std::min<uint64_t>((nw + 255) / 256, (uint64_t) ctx->num_cus * 16);
    hipLaunchKernelGGL(k_unpack2b, dim3(grid), dim3(256), 0, s, (const uint32_t *) packed_dev, n_bases, out_dev);
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

} // namespace kmu
