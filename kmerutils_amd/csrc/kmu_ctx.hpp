// kmu_ctx.hpp -- host-side context of libkmu: device, stream, error text, workspace arena, per-kernel timing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/kmu.h"
#ifndef KMU_DIAG
#define KMU_DIAG 0 // diagnostic builds only (KMU_BUILD_DEFS=-DKMU_DIAG=1): phase ablations / clocks, partition experiments
#endif

struct kmu_comm;
struct kmu_ctx {
    int device = 0;
    kmu_comm *comm = nullptr; // kmu_comm_init / kmu_comm_init_custom (kmu_comm.hip)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool async_device = false;
    bool err_unread = false; // the device error word holds bits no host read has seen yet (get_err_word / check_err_word)
    uint32_t lds_attr_set = 0; // kernels whose MaxDynamicSharedMemorySize was raised on THIS context's device (bit per kernel group)
    std::string err;
    // kmu_sketch_partial / kmu_sketch_hashed_partial: where the all-sequences paths leave their per-slot minima instead of
    // turning them into a signature (device memory; null in every other call)
    uint64_t *partial_out = nullptr;
    kmu_hll_params hll = {1.001, 20.0, 65534u, 0u}; // SetSketchParams of KMU_ALGO_HLL calls (kmu_set_hll_params)
    // grow-on-demand device scratch buffers, keyed by purpose (never shrunk; freed with the context)
    struct Buf {
        void *p = nullptr;
        size_t bytes = 0;
    };
    std::map<std::string, Buf> bufs;
    // pinned host staging
    std::map<std::string, Buf> hbufs;
    // profiling
    bool profiling = false;
    struct Pending {
        std::string name;
        hipEvent_t a, b;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    struct Stat {
        uint64_t launches = 0;
        double ms = 0;
    };
    std::map<std::string, Stat> stats;
    int num_cus = 256;
    size_t lds_per_block = 65536;
    // kmu_sketch_count on host buffers: uploads and downloads run on streams of their own, next to the kernels
    hipStream_t pipe_h2d = nullptr, pipe_d2h = nullptr;
};

namespace kmu {

extern thread_local std::string g_create_error;

inline int fail(kmu_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

#define KMU_HIP(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return kmu::fail((ctx), e_ == hipErrorOutOfMemory ? KMU_E_OOM : KMU_E_HIP, "%s: %s (%s:%d)", #expr, \
                             hipGetErrorString(e_), __FILE__, __LINE__);                                \
    } while (0)

#define KMU_TRY(expr)           \
    do {                        \
        int rc_ = (expr);       \
        if (rc_ != KMU_OK) return rc_; \
    } while (0)

// device scratch buffer of at least `bytes`
inline int dev_buf(kmu_ctx *ctx, const char *name, size_t bytes, void **out) {
    auto &b = ctx->bufs[name];
    if (b.bytes < bytes) {
        if (b.p) {
            KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
            KMU_HIP(ctx, hipFree(b.p));
            b.p = nullptr;
            b.bytes = 0;
        }
        size_t want = bytes + bytes / 8 + 256;
        KMU_HIP(ctx, hipMalloc(&b.p, want));
        b.bytes = want;
    }
    *out = b.p;
    return KMU_OK;
}
inline int host_buf(kmu_ctx *ctx, const char *name, size_t bytes, void **out) {
    auto &b = ctx->hbufs[name];
    if (b.bytes < bytes) {
        if (b.p) {
            KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
            KMU_HIP(ctx, hipHostFree(b.p));
            b.p = nullptr;
            b.bytes = 0;
        }
        size_t want = bytes + bytes / 8 + 256;
        KMU_HIP(ctx, hipHostMalloc(&b.p, want, hipHostMallocDefault));
        b.bytes = want;
    }
    *out = b.p;
    return KMU_OK;
}

// RAII-ish timing scope around one kernel launch (hipEvents on the context stream)
struct KernelTimer {
    kmu_ctx *ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char *name;
    KernelTimer(kmu_ctx *c, const char *n) : ctx(c), name(n) {
        if (!ctx->profiling) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!ctx->event_pool.empty()) {
                e = ctx->event_pool.back();
                ctx->event_pool.pop_back();
            } else {
                (void) hipEventCreate(&e);
            }
            return e;
        };
        a = get();
        b = get();
        (void) hipEventRecord(a, ctx->stream);
    }
    ~KernelTimer() {
        if (!ctx->profiling) return;
        (void) hipEventRecord(b, ctx->stream);
        ctx->pending.push_back({name, a, b});
    }
};

inline void profile_collect(kmu_ctx *ctx) {
    if (ctx->pending.empty()) return;
    (void) hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &s = ctx->stats[p.name];
            s.launches++;
            s.ms += ms;
        }
        ctx->event_pool.push_back(p.a);
        ctx->event_pool.push_back(p.b);
    }
    ctx->pending.clear();
}

// sequence inputs resolved to device memory
struct DevSeqs {
    const uint8_t *bases = nullptr;
    const uint64_t *offsets = nullptr;        // n_seq + 1
    const uint64_t *packed_offsets = nullptr; // n_seq (+1) or null
    uint64_t total_bytes = 0;                 // size of `bases` in bytes
    uint32_t n_seq = 0;
    int packed = 0;
    std::vector<uint64_t> h_offsets; // host copy of offsets (always available)
    std::vector<uint64_t> h_packed_offsets;
};

int stage_sequences(kmu_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, const uint64_t *packed_offsets,
                    uint32_t n_seq, int input_kind, int mem, DevSeqs *out);

int finish_call(kmu_ctx *ctx, int mem);

// device error word (bit flags set by kernels)
enum : uint32_t { DERR_NON_ACGT = 1u, DERR_TABLE_FULL = 2u, DERR_BAD_AA = 4u, DERR_EMPTY_SEQ = 8u, DERR_BAD_RANGE = 16u };
int get_err_word(kmu_ctx *ctx, uint32_t **out); // zeroed on the stream
int check_err_word(kmu_ctx *ctx, uint32_t *d_err);

void comm_free(kmu_ctx *ctx); // kmu_comm.hip
// kmu_count.hip, for kmu_sketch_count
int count_add_device_begin(kmu_counter *c, DevSeqs &ds, const uint64_t *host_offsets, int mem, uint32_t *d_err);
int count_add_device_end(kmu_counter *c);
kmu_ctx *counter_ctx(kmu_counter *c);
int count_chunked_begin(kmu_counter *c, DevSeqs &all, const uint64_t *host_offsets, uint32_t *d_err, void **handle, int *on);
int count_chunked_level1(kmu_counter *c, void *handle, uint64_t bases_ready);
int count_chunked_finish(kmu_counter *c, void *handle);
void count_chunked_abort(void *handle);

int check_kmer(kmu_ctx *ctx, int kmer_type, int k);
inline bool kmer_is_aa(int t) { return t == KMU_KMERAA32BIT || t == KMU_KMERAA64BIT; }
inline int kmer_val_bytes(int t) { return (t == KMU_KMER64BIT || t == KMU_KMERAA64BIT) ? 8 : 4; }
bool fhash_valid(int fhash, int kmer_type);

// exclusive scan of n u32 values into u64 offsets, out[n] = total (kmu_ingest.hip)
int device_scan_u32(kmu_ctx *ctx, const uint32_t *in, uint64_t n, uint64_t *out);

// radix partition of a device u64 array by the top bits of fmix64(key) (kmu_count.hip)
int partition_u64(kmu_ctx *ctx, const uint64_t *in, uint64_t n, int region_bits, const uint64_t **items_out,
                  const uint64_t **bounds_out, bool hashed_out = false);

} // namespace kmu
