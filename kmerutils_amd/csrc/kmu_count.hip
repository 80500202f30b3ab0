// kmu_count.hip -- k-mer counting on gfx950 behind the KmerCountT contract.
//
// Reference (src/base/kmercount.rs:241-277): a cuckoo filter holds k-mers seen once, a counting Bloom filter
// the counts >= 2; both are randomised per process, so the observable contract is "exact multiplicity of the
// canonical k-mer, reported saturated at 2^bits - 1" (plus ~3 % false positives the reference itself treats as
// noise).  Here: one exact open-addressing table in HBM (8-byte canonical value + 4-byte count per slot), updated
// with one 64-bit CAS + one 32-bit atomic add per k-mer occurrence.  The bases are consumed as ONE flat stream of
// aligned 16-byte words (perfect coalescing and load balance whatever the read lengths); each lane finds the
// read that owns its word by binary search in `offsets` and masks the k-mers that would straddle a read end.
#include <algorithm>
#include <vector>

#include "kmu_ctx.hpp"
#include "kmu_stream.h"

struct kmu_counter {
    kmu_ctx *ctx = nullptr;
    kmu_count_params p{};
    uint64_t nslots = 0; // power of two
    int lg = 0;
    uint64_t *keys = nullptr;
    uint32_t *counts = nullptr;
    uint64_t *scalars = nullptr; // device: [0] distinct, [1] unique, [2] cursor
};

namespace kmu {

static constexpr uint64_t CKEY_EMPTY = 0xFFFFFFFFFFFFFFFFull; // canonical values are < 2^62

__device__ __forceinline__ uint64_t fmix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

struct CountTable {
    uint64_t *keys;
    uint32_t *counts;
    uint64_t mask;
    int shift; // 64 - lg
};

// KmerCounter::insert_kmer (kmercount.rs:241-267), exact form: count[v] += add
__device__ __forceinline__ bool count_insert(const CountTable &t, uint64_t v, uint32_t add) {
    uint64_t idx = fmix64(v) >> t.shift;
    for (uint64_t probes = 0; probes <= t.mask; probes++) {
        uint64_t cur = __hip_atomic_load(&t.keys[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == CKEY_EMPTY) {
            cur = atomicCAS((unsigned long long *) &t.keys[idx], (unsigned long long) CKEY_EMPTY, (unsigned long long) v);
            if (cur == CKEY_EMPTY) cur = v;
        }
        if (cur == v) {
            atomicAdd(&t.counts[idx], add);
            return true;
        }
        idx = (idx + 1) & t.mask;
    }
    return false;
}

__device__ __forceinline__ uint32_t count_lookup(const CountTable &t, uint64_t v) {
    uint64_t idx = fmix64(v) >> t.shift;
    for (uint64_t probes = 0; probes <= t.mask; probes++) {
        uint64_t cur = t.keys[idx];
        if (cur == v) return t.counts[idx];
        if (cur == CKEY_EMPTY) return 0;
        idx = (idx + 1) & t.mask;
    }
    return 0;
}

// largest i with offsets[i] <= g  (offsets[0] = 0, offsets[n] = total > g)
__device__ __forceinline__ uint32_t find_read(const uint64_t *offsets, uint32_t n, uint64_t g) {
    uint32_t lo = 0, hi = n; // invariant offsets[lo] <= g < offsets[hi]
    while (hi - lo > 1) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (offsets[mid] <= g) lo = mid;
        else hi = mid;
    }
    return lo;
}

// flat stream over all bases (ASCII input)
__global__ void __launch_bounds__(256) k_count_add_flat(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq,
                                                        int k, CountTable t, uint32_t *err) {
    const uint64_t total = offsets[n_seq];
    const uint64_t nwords = (total + 15) / 16;
    const uint64_t nsteps = (nwords + 63) / 64;
    const uint64_t wave_global = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves_global = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    SeqView s;
    s.base = bases; s.begin = 0; s.len = total; s.total = total; s.packed = 0;
    const int lane = lane_id();
    const int sh = 64 - 2 * k;
    uint32_t anybad = 0, full = 0;
    for (uint64_t st = wave_global; st < nsteps; st += nwaves_global) {
        const uint64_t widx = st * 64 + lane;
        uint32_t bad, bad2;
        uint32_t w0 = load_code_word(s, widx, bad);
        uint32_t ex = load_code_word(s, st * 64 + 64 + (uint64_t) (lane & 1), bad2);
        uint32_t e0 = bcast_u32(ex, 0), e1 = bcast_u32(ex, 1);
        uint32_t w1 = shfl_down_u32(w0, 1), w2 = shfl_down_u32(w0, 2);
        if (lane == 63) { w1 = e0; w2 = e1; }
        if (lane == 62) { w2 = e0; }
        anybad |= bad;
        const uint64_t g0 = widx * 16;
        if (g0 >= total) continue;
        const uint64_t hi = ((uint64_t) w0 << 32) | w1;
        uint32_t r = find_read(offsets, n_seq, g0);
        uint64_t rend = offsets[r + 1];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint64_t g = g0 + j;
            while (g >= rend && r + 1 < n_seq) { r++; rend = offsets[r + 1]; }
            if (g + k <= rend) {
                uint64_t v = (hi << (2 * j)) | (((uint64_t) w2 << (2 * j)) >> 32);
                uint64_t val = v >> sh;
                uint64_t rc = revcomp_val(val, k);
                uint64_t canon = rc < val ? rc : val; // kmer.reverse_complement().min(kmer), kmercount.rs:938
                if (!count_insert(t, canon, 1u)) full = 1;
            }
        }
    }
    if (anybad) atomicOr(err, DERR_NON_ACGT);
    if (full) atomicOr(err, DERR_TABLE_FULL);
}

// per-read form (packed input): one wave per read step
__global__ void __launch_bounds__(256) k_count_add_reads(const uint8_t *bases, const uint64_t *offsets,
                                                         const uint64_t *packed_offsets, uint32_t n_seq, int packed,
                                                         uint64_t total_bytes, int k, CountTable t, uint32_t *err) {
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t bad = 0, full = 0;
    for (uint32_t i = blockIdx.x; i < n_seq; i += gridDim.x) {
        SeqView s;
        s.base = bases;
        s.len = offsets[i + 1] - offsets[i];
        s.packed = packed;
        if (packed) {
            s.begin = packed_offsets[i];
            s.total = total_bytes ? total_bytes : (packed_offsets[n_seq - 1] + (offsets[n_seq] - offsets[n_seq - 1] + 3) / 4);
        } else {
            s.begin = offsets[i];
            s.total = total_bytes ? total_bytes : offsets[n_seq];
        }
        const uint64_t nk = s.len >= (uint64_t) k ? s.len - k + 1 : 0;
        if (nk == 0) { bad |= wave_validate_seq(s, wave, nwaves, false); continue; }
        const uint64_t nsteps = (seq_num_words(s) + 63) / 64;
        for (uint64_t st = wave; st < nsteps; st += nwaves)
            bad |= wave_step_kmers(s, k, st, 0, nk, [&](uint64_t, uint64_t val, uint64_t rc) {
                uint64_t canon = rc < val ? rc : val;
                if (!count_insert(t, canon, 1u)) full = 1;
            });
    }
    if (bad) atomicOr(err, DERR_NON_ACGT);
    if (full) atomicOr(err, DERR_TABLE_FULL);
}

__global__ void __launch_bounds__(256) k_count_add_kmers(const uint64_t *kmers, const uint32_t *adds, uint64_t n,
                                                         CountTable t, uint32_t *err) {
    uint32_t full = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        if (!count_insert(t, kmers[i], adds ? adds[i] : 1u)) full = 1;
    if (full) atomicOr(err, DERR_TABLE_FULL);
}

__global__ void __launch_bounds__(256) k_count_query(const uint64_t *kmers, uint64_t n, CountTable t, uint32_t maxc,
                                                     uint32_t *out) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        uint32_t c = count_lookup(t, kmers[i]);
        out[i] = c > maxc ? maxc : c;
    }
}

// [0] += occupied slots, [1] += slots with count == 1, [3] += slots with count >= min_count in partition
__global__ void __launch_bounds__(256) k_count_stats(CountTable t, uint64_t *scalars) {
    uint64_t d = 0, u = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= t.mask; i += (uint64_t) gridDim.x * blockDim.x) {
        if (t.keys[i] != CKEY_EMPTY) {
            d++;
            u += t.counts[i] == 1u;
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        d += ((uint64_t) (uint32_t) __shfl_xor((int) (d >> 32), o, 64) << 32) | (uint32_t) __shfl_xor((int) (uint32_t) d, o, 64);
        u += ((uint64_t) (uint32_t) __shfl_xor((int) (u >> 32), o, 64) << 32) | (uint32_t) __shfl_xor((int) (uint32_t) u, o, 64);
    }
    if (lane_id() == 0) {
        if (d) atomicAdd((unsigned long long *) &scalars[0], (unsigned long long) d);
        if (u) atomicAdd((unsigned long long *) &scalars[1], (unsigned long long) u);
    }
}

// owner of a k-mer in an n-way key partition: DispatchableT, kmercount.rs:382-420
__device__ __forceinline__ uint32_t kmer_owner(uint64_t v, int w32, uint32_t n_parts) {
    return w32 ? (uint32_t) (int32_hash((uint32_t) v) % n_parts) : (uint32_t) (int64_hash(v) % (uint64_t) n_parts);
}

// compact (kmer, count) with count >= min_count (and owner == part when n_parts > 0); cap-limited
__global__ void __launch_bounds__(256) k_count_select(CountTable t, uint32_t min_count, uint32_t maxc, int w32,
                                                      uint32_t part, uint32_t n_parts, uint64_t cap, uint64_t *kmers,
                                                      uint32_t *counts, uint64_t *cursor) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= t.mask; i += (uint64_t) gridDim.x * blockDim.x) {
        uint64_t key = t.keys[i];
        if (key == CKEY_EMPTY) continue;
        uint32_t c = t.counts[i];
        if (c < min_count) continue;
        if (n_parts && kmer_owner(key, w32, n_parts) != part) continue;
        uint64_t pos = atomicAdd((unsigned long long *) cursor, 1ull);
        if (kmers && pos < cap) {
            kmers[pos] = key;
            counts[pos] = c > maxc ? maxc : c;
        }
    }
}

} // namespace kmu

using namespace kmu;

static CountTable table_of(const kmu_counter *c) {
    CountTable t;
    t.keys = c->keys;
    t.counts = c->counts;
    t.mask = c->nslots - 1;
    t.shift = 64 - c->lg;
    return t;
}
static uint32_t max_count(const kmu_counter *c) { return c->p.counter_bits == 8 ? 255u : 65535u; }
static int grid_for(const kmu_ctx *ctx, uint64_t n, int per_block) {
    uint64_t blocks = (n + per_block - 1) / per_block;
    uint64_t cap = (uint64_t) ctx->num_cus * 8;
    return (int) std::max<uint64_t>(1, std::min(blocks, cap));
}

extern "C" {

int kmu_count_reset(kmu_counter *c) {
    if (!c) return KMU_E_BAD_ARG;
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    KMU_HIP(ctx, hipMemsetAsync(c->keys, 0xFF, c->nslots * 8, ctx->stream));
    KMU_HIP(ctx, hipMemsetAsync(c->counts, 0, c->nslots * 4, ctx->stream));
    return KMU_OK;
}

int kmu_count_create(kmu_ctx *ctx, const kmu_count_params *p, kmu_counter **out) {
    if (!ctx || !p || !out) return KMU_E_BAD_ARG;
    *out = nullptr;
    KMU_TRY(check_kmer(ctx, p->kmer_type, p->kmer_size));
    if (kmer_is_aa(p->kmer_type)) return fail(ctx, KMU_E_BAD_ARG, "counting is defined on DNA k-mers (canonical form)");
    if (p->counter_bits != 8 && p->counter_bits != 16) return fail(ctx, KMU_E_BAD_ARG, "counter_bits must be 8 or 16");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    kmu_counter *c = new kmu_counter();
    c->ctx = ctx;
    c->p = *p;
    uint64_t want = std::max<uint64_t>(1024, p->capacity_hint + p->capacity_hint / 2); // load factor <= 2/3
    c->lg = 10;
    while ((1ull << c->lg) < want) c->lg++;
    c->nslots = 1ull << c->lg;
    hipError_t e1 = hipMalloc((void **) &c->keys, c->nslots * 8);
    hipError_t e2 = e1 == hipSuccess ? hipMalloc((void **) &c->counts, c->nslots * 4) : e1;
    hipError_t e3 = e2 == hipSuccess ? hipMalloc((void **) &c->scalars, 64) : e2;
    if (e3 != hipSuccess) {
        if (c->keys) (void) hipFree(c->keys);
        if (c->counts) (void) hipFree(c->counts);
        delete c;
        (void) hipGetLastError();
        return fail(ctx, KMU_E_OOM, "cannot allocate a %llu-slot count table", (unsigned long long) (1ull << c->lg));
    }
    int rc = kmu_count_reset(c);
    if (rc) { kmu_count_destroy(c); return rc; }
    *out = c;
    return KMU_OK;
}

void kmu_count_destroy(kmu_counter *c) {
    if (!c) return;
    (void) hipSetDevice(c->ctx->device);
    (void) hipStreamSynchronize(c->ctx->stream);
    if (c->keys) (void) hipFree(c->keys);
    if (c->counts) (void) hipFree(c->counts);
    if (c->scalars) (void) hipFree(c->scalars);
    delete c;
}

int kmu_count_add_reads(kmu_counter *c, const uint8_t *bases, const uint64_t *offsets, const uint64_t *packed_offsets,
                        uint32_t n_seq, int input_kind, int mem) {
    if (!c) return KMU_E_BAD_ARG;
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, packed_offsets, n_seq, input_kind, mem, &ds));
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    if (n_seq) {
        CountTable t = table_of(c);
        if (!ds.packed) {
            int grid = ctx->num_cus * 8;
            KernelTimer tm(ctx, "k_count_add_flat");
            hipLaunchKernelGGL(k_count_add_flat, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets, n_seq,
                               c->p.kmer_size, t, d_err);
        } else {
            int grid = (int) std::min<uint64_t>(n_seq, (uint64_t) ctx->num_cus * 8);
            KernelTimer tm(ctx, "k_count_add_reads");
            hipLaunchKernelGGL(k_count_add_reads, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets,
                               ds.packed_offsets, n_seq, ds.packed, ds.total_bytes, c->p.kmer_size, t, d_err);
        }
        KMU_HIP(ctx, hipGetLastError());
    }
    if (!(mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
    return finish_call(ctx, mem);
}

static int add_entries(kmu_counter *c, const uint64_t *kmers, const uint32_t *counts, uint64_t n, int mem) {
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) return KMU_OK;
    const uint64_t *d_k = kmers;
    const uint32_t *d_c = counts;
    if (mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "cnt.in.k", n * 8, &q));
        KMU_HIP(ctx, hipMemcpyAsync(q, kmers, n * 8, hipMemcpyHostToDevice, ctx->stream));
        d_k = (const uint64_t *) q;
        if (counts) {
            KMU_TRY(dev_buf(ctx, "cnt.in.c", n * 4, &q));
            KMU_HIP(ctx, hipMemcpyAsync(q, counts, n * 4, hipMemcpyHostToDevice, ctx->stream));
            d_c = (const uint32_t *) q;
        }
    }
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    {
        KernelTimer tm(ctx, "k_count_add_kmers");
        hipLaunchKernelGGL(k_count_add_kmers, dim3(grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, d_k, d_c, n,
                           table_of(c), d_err);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (!(mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
    return finish_call(ctx, mem);
}

int kmu_count_add_kmers(kmu_counter *c, const uint64_t *canon_kmers, uint64_t n, int mem) {
    if (!c || (!canon_kmers && n)) return KMU_E_BAD_ARG;
    return add_entries(c, canon_kmers, nullptr, n, mem);
}

int kmu_count_merge_entries(kmu_counter *c, const uint64_t *kmers, const uint32_t *counts, uint64_t n, int mem) {
    if (!c || ((!kmers || !counts) && n)) return KMU_E_BAD_ARG;
    return add_entries(c, kmers, counts, n, mem);
}

int kmu_count_query(kmu_counter *c, const uint64_t *canon_kmers, uint64_t n, int mem, uint32_t *counts_out) {
    if (!c || ((!canon_kmers || !counts_out) && n)) return KMU_E_BAD_ARG;
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) return KMU_OK;
    const uint64_t *d_k = canon_kmers;
    uint32_t *d_o = counts_out;
    if (mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "cnt.in.k", n * 8, &q));
        KMU_HIP(ctx, hipMemcpyAsync(q, canon_kmers, n * 8, hipMemcpyHostToDevice, ctx->stream));
        d_k = (const uint64_t *) q;
        KMU_TRY(dev_buf(ctx, "cnt.out.c", n * 4, &q));
        d_o = (uint32_t *) q;
    }
    {
        KernelTimer tm(ctx, "k_count_query");
        hipLaunchKernelGGL(k_count_query, dim3(grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, d_k, n, table_of(c),
                           max_count(c), d_o);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (mem == KMU_MEM_HOST) KMU_HIP(ctx, hipMemcpyAsync(counts_out, d_o, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    return finish_call(ctx, mem);
}

static int count_stats(kmu_counter *c, uint64_t *distinct, uint64_t *unique) {
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    KMU_HIP(ctx, hipMemsetAsync(c->scalars, 0, 64, ctx->stream));
    {
        KernelTimer tm(ctx, "k_count_stats");
        hipLaunchKernelGGL(k_count_stats, dim3(grid_for(ctx, c->nslots, 1024)), dim3(256), 0, ctx->stream, table_of(c),
                           c->scalars);
    }
    KMU_HIP(ctx, hipGetLastError());
    uint64_t h[2];
    KMU_HIP(ctx, hipMemcpyAsync(h, c->scalars, 16, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (distinct) *distinct = h[0];
    if (unique) *unique = h[1];
    return KMU_OK;
}

int kmu_count_nb_distinct(kmu_counter *c, uint64_t *out) {
    if (!c || !out) return KMU_E_BAD_ARG;
    return count_stats(c, out, nullptr);
}
int kmu_count_nb_unique(kmu_counter *c, uint64_t *out) {
    if (!c || !out) return KMU_E_BAD_ARG;
    return count_stats(c, nullptr, out);
}

// shared by dump / export: select into device buffers, then hand over
static int select_entries(kmu_counter *c, uint32_t min_count, uint32_t part, uint32_t n_parts, uint64_t *kmers_out,
                          uint32_t *counts_out, uint64_t cap, int mem, bool sort, uint64_t *n_out) {
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    const int w32 = kmer_val_bytes(c->p.kmer_type) == 4;
    uint64_t *d_k = nullptr;
    uint32_t *d_c = nullptr;
    if (kmers_out) {
        if (mem == KMU_MEM_HOST) {
            void *q;
            KMU_TRY(dev_buf(ctx, "cnt.sel.k", cap * 8 + 8, &q));
            d_k = (uint64_t *) q;
            KMU_TRY(dev_buf(ctx, "cnt.sel.c", cap * 4 + 8, &q));
            d_c = (uint32_t *) q;
        } else {
            d_k = kmers_out;
            d_c = counts_out;
        }
    }
    KMU_HIP(ctx, hipMemsetAsync(c->scalars, 0, 64, ctx->stream));
    {
        KernelTimer tm(ctx, "k_count_select");
        hipLaunchKernelGGL(k_count_select, dim3(grid_for(ctx, c->nslots, 1024)), dim3(256), 0, ctx->stream, table_of(c),
                           min_count, max_count(c), w32, part, n_parts, cap, d_k, d_c, c->scalars + 2);
    }
    KMU_HIP(ctx, hipGetLastError());
    uint64_t n = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&n, c->scalars + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    if (!kmers_out) return KMU_OK;
    if (n > cap) return fail(ctx, KMU_E_BAD_ARG, "output capacity %llu < %llu records", (unsigned long long) cap, (unsigned long long) n);
    if (mem == KMU_MEM_HOST) {
        KMU_HIP(ctx, hipMemcpy(kmers_out, d_k, n * 8, hipMemcpyDeviceToHost));
        KMU_HIP(ctx, hipMemcpy(counts_out, d_c, n * 4, hipMemcpyDeviceToHost));
        if (sort) {
            std::vector<uint64_t> idx(n);
            for (uint64_t i = 0; i < n; i++) idx[i] = i;
            std::sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return kmers_out[a] < kmers_out[b]; });
            std::vector<uint64_t> k2(n);
            std::vector<uint32_t> c2(n);
            for (uint64_t i = 0; i < n; i++) { k2[i] = kmers_out[idx[i]]; c2[i] = counts_out[idx[i]]; }
            memcpy(kmers_out, k2.data(), n * 8);
            memcpy(counts_out, c2.data(), n * 4);
        }
    }
    return KMU_OK;
}

int kmu_count_dump(kmu_counter *c, uint32_t min_count, uint64_t *kmers_out, uint32_t *counts_out, uint64_t cap,
                   uint64_t *n_out) {
    if (!c || !n_out || (kmers_out && !counts_out)) return KMU_E_BAD_ARG;
    return select_entries(c, min_count, 0, 0, kmers_out, counts_out, cap, KMU_MEM_HOST, true, n_out);
}

int kmu_count_export_part(kmu_counter *c, uint32_t part, uint32_t n_parts, uint64_t *kmers_out, uint32_t *counts_out,
                          uint64_t cap, int mem, uint64_t *n_out) {
    if (!c || !n_out || n_parts == 0 || part >= n_parts || (kmers_out && !counts_out)) return KMU_E_BAD_ARG;
    // raw (unclamped) counts are exported so that merged totals saturate only once, at query time
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    const int w32 = kmer_val_bytes(c->p.kmer_type) == 4;
    uint64_t *d_k = nullptr;
    uint32_t *d_c = nullptr;
    if (kmers_out) {
        if (mem == KMU_MEM_HOST) {
            void *q;
            KMU_TRY(dev_buf(ctx, "cnt.sel.k", cap * 8 + 8, &q));
            d_k = (uint64_t *) q;
            KMU_TRY(dev_buf(ctx, "cnt.sel.c", cap * 4 + 8, &q));
            d_c = (uint32_t *) q;
        } else {
            d_k = kmers_out;
            d_c = counts_out;
        }
    }
    KMU_HIP(ctx, hipMemsetAsync(c->scalars, 0, 64, ctx->stream));
    {
        KernelTimer tm(ctx, "k_count_select");
        hipLaunchKernelGGL(k_count_select, dim3(grid_for(ctx, c->nslots, 1024)), dim3(256), 0, ctx->stream, table_of(c),
                           1u, 0xFFFFFFFFu, w32, part, n_parts, cap, d_k, d_c, c->scalars + 2);
    }
    KMU_HIP(ctx, hipGetLastError());
    uint64_t n = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&n, c->scalars + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    if (!kmers_out) return KMU_OK;
    if (n > cap) return fail(ctx, KMU_E_BAD_ARG, "output capacity %llu < %llu records", (unsigned long long) cap, (unsigned long long) n);
    if (mem == KMU_MEM_HOST) {
        KMU_HIP(ctx, hipMemcpy(kmers_out, d_k, n * 8, hipMemcpyDeviceToHost));
        KMU_HIP(ctx, hipMemcpy(counts_out, d_c, n * 4, hipMemcpyDeviceToHost));
    }
    return KMU_OK;
}

int kmu_count_retain_part(kmu_counter *c, uint32_t part, uint32_t n_parts) {
    if (!c || n_parts == 0 || part >= n_parts) return KMU_E_BAD_ARG;
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t n = 0;
    KMU_TRY(kmu_count_export_part(c, part, n_parts, nullptr, nullptr, 0, KMU_MEM_DEVICE, &n));
    void *k = nullptr, *cc = nullptr;
    KMU_TRY(dev_buf(ctx, "cnt.keep.k", n * 8 + 8, &k));
    KMU_TRY(dev_buf(ctx, "cnt.keep.c", n * 4 + 8, &cc));
    uint64_t n2 = 0;
    KMU_TRY(kmu_count_export_part(c, part, n_parts, (uint64_t *) k, (uint32_t *) cc, n, KMU_MEM_DEVICE, &n2));
    KMU_TRY(kmu_count_reset(c));
    int rc = add_entries(c, (const uint64_t *) k, (const uint32_t *) cc, n2, KMU_MEM_DEVICE);
    if (rc) return rc;
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KMU_OK;
}

} // extern "C"
