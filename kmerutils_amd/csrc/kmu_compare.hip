// kmu_compare.hip -- what the reference does with the signatures right after the path.
//
//  * slot-equality between signature rows: probminhash_get_jaccard_objects (src/sketching/seqsketchjaccard.rs:86-108),
//    DistBlockSketched / distance_jaccard_serial (src/sketching/seqblocksketch.rs:419-440), DistHamming on u32 rows
//    (src/bin/datasketcher.rs:156-185).  The kernels return the number of EQUAL slots; jaccard = eq / m,
//    block distance = (m - eq) / m (1.0 when both blocks come from the same sequence: a host-side rule).
//  * bottom-k: minhash_distance / mininvhash_distance (src/sketching/minhash.rs:134-190, 295-340): the merge walk over
//    two ascending hash rows, with the reference's way of topping up `total`.
//
// Rows are compared as raw 4- or 8-byte words (u32 / u64 signatures, f32 / f64 SuperMinHash values alike).
#include <algorithm>

#include "kmu_ctx.hpp"
#include "kmu_device.h"

namespace kmu {

// one wave per pair: 64 slots per step, ballot + popcount
template <typename W>
__global__ void __launch_bounds__(256) k_sig_equal_pairs(const W *a, const W *b, uint32_t m, const uint32_t *ia,
                                                         const uint32_t *ib, uint64_t n_pairs, uint32_t *out) {
    const uint64_t wave_global = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = (uint32_t) lane_id();
    for (uint64_t p = wave_global; p < n_pairs; p += nwaves) {
        const W *ra = a + (uint64_t) ia[p] * m, *rb = b + (uint64_t) ib[p] * m;
        uint32_t eq = 0;
        for (uint32_t t0 = 0; t0 < m; t0 += 64) { // uniform trip count
            const uint32_t t = t0 + lane;
            const bool e = t < m && ra[t] == rb[t];
            eq += (uint32_t) __popcll(__ballot(e));
        }
        if (lane == 0) out[p] = eq;
    }
}

// all pairs of na x nb rows.  A workgroup owns a 64 x 64 tile of pairs; the slots go through LDS in chunks of MC, every
// thread keeps a 4 x 4 block of counters (rows i0 + ty + 16 r, columns j0 + tx + 16 c).
static constexpr int MT = 64, MC = 32;
template <typename W>
__global__ void __launch_bounds__(256) k_sig_equal_matrix(const W *a, uint32_t na, const W *b, uint32_t nb, uint32_t m,
                                                          uint16_t *out) {
    __shared__ W la[MT][MC + 1], lb[MT][MC + 1];
    const uint32_t tilesx = (nb + MT - 1) / MT;
    const uint32_t i0 = (blockIdx.x / tilesx) * MT, j0 = (blockIdx.x % tilesx) * MT;
    const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    uint32_t acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[r][c] = 0;
    for (uint32_t t0 = 0; t0 < m; t0 += MC) {
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < (uint32_t) MT * MC; e += blockDim.x) {
            const uint32_t row = e / MC, t = e % MC;
            // slots past m are filled with values that differ between the two sides
            la[row][t] = (i0 + row < na && t0 + t < m) ? a[(uint64_t) (i0 + row) * m + t0 + t] : (W) 0;
            lb[row][t] = (j0 + row < nb && t0 + t < m) ? b[(uint64_t) (j0 + row) * m + t0 + t] : (W) 1;
        }
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < MC; t++) {
            W va[4], vb[4];
#pragma unroll
            for (int r = 0; r < 4; r++) va[r] = la[ty + 16 * r][t];
#pragma unroll
            for (int c = 0; c < 4; c++) vb[c] = lb[tx + 16 * c][t];
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[r][c] += va[r] == vb[c];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t i = i0 + ty + 16 * r, j = j0 + tx + 16 * c;
            if (i < na && j < nb) out[(uint64_t) i * nb + j] = (uint16_t) acc[r][c];
        }
}

// minhash_distance: one thread per pair.  Rows hold ascending hashes padded with u64::MAX (kmu_sketch, KMU_ALGO_BOTTOMK).
// out[3p..3p+2] = common, total, i (the number of items of the first row walked: containment = common / i)
__global__ void __launch_bounds__(256) k_minhash_distance(const uint64_t *a, const uint64_t *b, uint32_t m, const uint32_t *ia,
                                                          const uint32_t *ib, uint64_t n_pairs, uint32_t *out) {
    for (uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += (uint64_t) gridDim.x * blockDim.x) {
        const uint64_t *r1 = a + (uint64_t) ia[p] * m, *r2 = b + (uint64_t) ib[p] * m;
        uint32_t n1 = 0, n2 = 0;
        while (n1 < m && r1[n1] != 0xFFFFFFFFFFFFFFFFull) n1++;
        while (n2 < m && r2[n2] != 0xFFFFFFFFFFFFFFFFull) n2++;
        uint32_t i = 0, j = 0, common = 0, total = 0;
        while (i < n1 && j < n2) { // minhash.rs:150-164
            const uint64_t x = r1[i], y = r2[j];
            if (x < y) i++;
            else if (y < x) j++;
            else { i++; j++; common++; }
            total++;
            if (total >= n1) break;
        }
        if (total < n1) { // minhash.rs:168-180 -- both top-ups are measured against the FIRST sketch's length, as upstream
            if (i < n1) total += n1 - i;
            if (j < n1) total += n1 - j;
            if (total > n1) total = n1;
        }
        out[3 * p] = common;
        out[3 * p + 1] = total;
        out[3 * p + 2] = i;
    }
}

// stage `bytes` of a host buffer (or pass a device pointer through)
static int to_device(kmu_ctx *ctx, const char *name, const void *p, size_t bytes, int mem, const void **out) {
    if (mem == KMU_MEM_DEVICE || !p) { *out = p; return KMU_OK; }
    void *d;
    KMU_TRY(dev_buf(ctx, name, bytes ? bytes : 1, &d));
    if (bytes) KMU_HIP(ctx, hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    *out = d;
    return KMU_OK;
}

} // namespace kmu

using namespace kmu;

static int sig_word_bytes(int sig_type) { return (sig_type == KMU_SIG_U32 || sig_type == KMU_SIG_F32) ? 4 : 8; }

extern "C" {

int kmu_sig_equal_pairs(kmu_ctx *ctx, const void *sig_a, uint32_t na, const void *sig_b, uint32_t nb, uint32_t m,
                        int sig_type, const uint32_t *ia, const uint32_t *ib, uint64_t n_pairs, int mem, uint32_t *out) {
    if (!ctx || !sig_a || !sig_b || !out || m == 0 || (n_pairs && (!ia || !ib))) return fail(ctx, KMU_E_BAD_ARG, "null argument or m == 0");
    if (sig_type < KMU_SIG_U32 || sig_type > KMU_SIG_F64) return fail(ctx, KMU_E_BAD_ARG, "bad sig_type %d", sig_type);
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (n_pairs == 0) return KMU_OK;
    const int wb = sig_word_bytes(sig_type);
    if (mem == KMU_MEM_HOST) // row indices are checked where the host can see them
        for (uint64_t p = 0; p < n_pairs; p++)
            if (ia[p] >= na || ib[p] >= nb) return fail(ctx, KMU_E_BAD_ARG, "pair %llu names a row out of range", (unsigned long long) p);
    const void *da, *db, *dia, *dib;
    KMU_TRY(to_device(ctx, "cmp.a", sig_a, (size_t) na * m * wb, mem, &da));
    KMU_TRY(to_device(ctx, "cmp.b", sig_b, (size_t) nb * m * wb, mem, &db));
    KMU_TRY(to_device(ctx, "cmp.ia", ia, n_pairs * 4, mem, &dia));
    KMU_TRY(to_device(ctx, "cmp.ib", ib, n_pairs * 4, mem, &dib));
    uint32_t *dout = out;
    if (mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "cmp.out", n_pairs * 4, &q));
        dout = (uint32_t *) q;
    }
    const int grid = (int) std::min<uint64_t>((n_pairs + 3) / 4, (uint64_t) ctx->num_cus * 16);
    {
        KernelTimer t(ctx, "k_sig_equal_pairs");
        if (wb == 4)
            hipLaunchKernelGGL(k_sig_equal_pairs<uint32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t *) da,
                               (const uint32_t *) db, m, (const uint32_t *) dia, (const uint32_t *) dib, n_pairs, dout);
        else
            hipLaunchKernelGGL(k_sig_equal_pairs<uint64_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint64_t *) da,
                               (const uint64_t *) db, m, (const uint32_t *) dia, (const uint32_t *) dib, n_pairs, dout);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (mem == KMU_MEM_HOST) KMU_HIP(ctx, hipMemcpyAsync(out, dout, n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    return finish_call(ctx, mem);
}

int kmu_sig_equal_matrix(kmu_ctx *ctx, const void *sig_a, uint32_t na, const void *sig_b, uint32_t nb, uint32_t m,
                         int sig_type, int mem, uint16_t *out) {
    if (!ctx || !sig_a || !sig_b || !out || m == 0) return fail(ctx, KMU_E_BAD_ARG, "null argument or m == 0");
    if (m > 65535u) return fail(ctx, KMU_E_UNSUPPORTED, "sketch_size %u does not fit the 16-bit counts of the matrix", m);
    if (sig_type < KMU_SIG_U32 || sig_type > KMU_SIG_F64) return fail(ctx, KMU_E_BAD_ARG, "bad sig_type %d", sig_type);
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (na == 0 || nb == 0) return KMU_OK;
    const int wb = sig_word_bytes(sig_type);
    const void *da, *db;
    KMU_TRY(to_device(ctx, "cmp.a", sig_a, (size_t) na * m * wb, mem, &da));
    KMU_TRY(to_device(ctx, "cmp.b", sig_b, (size_t) nb * m * wb, mem, &db));
    uint16_t *dout = out;
    if (mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "cmp.out", (size_t) na * nb * 2, &q));
        dout = (uint16_t *) q;
    }
    const uint64_t tiles = (uint64_t) ((na + MT - 1) / MT) * ((nb + MT - 1) / MT);
    if (tiles > 0x7FFFFFFFull) return fail(ctx, KMU_E_UNSUPPORTED, "matrix too large for one launch: split the rows");
    {
        KernelTimer t(ctx, "k_sig_equal_matrix");
        if (wb == 4)
            hipLaunchKernelGGL(k_sig_equal_matrix<uint32_t>, dim3((uint32_t) tiles), dim3(256), 0, ctx->stream,
                               (const uint32_t *) da, na, (const uint32_t *) db, nb, m, dout);
        else
            hipLaunchKernelGGL(k_sig_equal_matrix<uint64_t>, dim3((uint32_t) tiles), dim3(256), 0, ctx->stream,
                               (const uint64_t *) da, na, (const uint64_t *) db, nb, m, dout);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (mem == KMU_MEM_HOST)
        KMU_HIP(ctx, hipMemcpyAsync(out, dout, (size_t) na * nb * 2, hipMemcpyDeviceToHost, ctx->stream));
    return finish_call(ctx, mem);
}

int kmu_minhash_distance_pairs(kmu_ctx *ctx, const uint64_t *hashes_a, uint32_t na, const uint64_t *hashes_b, uint32_t nb,
                               uint32_t m, const uint32_t *ia, const uint32_t *ib, uint64_t n_pairs, int mem,
                               uint32_t *out) {
    if (!ctx || !hashes_a || !hashes_b || !out || m == 0 || (n_pairs && (!ia || !ib))) return fail(ctx, KMU_E_BAD_ARG, "null argument or m == 0");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (n_pairs == 0) return KMU_OK;
    if (mem == KMU_MEM_HOST)
        for (uint64_t p = 0; p < n_pairs; p++)
            if (ia[p] >= na || ib[p] >= nb) return fail(ctx, KMU_E_BAD_ARG, "pair %llu names a row out of range", (unsigned long long) p);
    const void *da, *db, *dia, *dib;
    KMU_TRY(to_device(ctx, "cmp.a", hashes_a, (size_t) na * m * 8, mem, &da));
    KMU_TRY(to_device(ctx, "cmp.b", hashes_b, (size_t) nb * m * 8, mem, &db));
    KMU_TRY(to_device(ctx, "cmp.ia", ia, n_pairs * 4, mem, &dia));
    KMU_TRY(to_device(ctx, "cmp.ib", ib, n_pairs * 4, mem, &dib));
    uint32_t *dout = out;
    if (mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "cmp.out", n_pairs * 12, &q));
        dout = (uint32_t *) q;
    }
    const int grid = (int) std::min<uint64_t>((n_pairs + 255) / 256, (uint64_t) ctx->num_cus * 8);
    {
        KernelTimer t(ctx, "k_minhash_distance");
        hipLaunchKernelGGL(k_minhash_distance, dim3(grid), dim3(256), 0, ctx->stream, (const uint64_t *) da,
                           (const uint64_t *) db, m, (const uint32_t *) dia, (const uint32_t *) dib, n_pairs, dout);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (mem == KMU_MEM_HOST) KMU_HIP(ctx, hipMemcpyAsync(out, dout, n_pairs * 12, hipMemcpyDeviceToHost, ctx->stream));
    return finish_call(ctx, mem);
}

} // extern "C"
