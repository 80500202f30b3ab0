// kmu_api.hip -- context management + L0/L1/L2 entry points of the C-ABI (include/kmu.h).
//
// gfx950 only.  No CPU fallback: without a HIP device kmu_create fails with KMU_E_NO_DEVICE.
#include <algorithm>

#include "kmu_ctx.hpp"
#include "kmu_stream.h"

namespace kmu {

thread_local std::string g_create_error;

int check_kmer(kmu_ctx *ctx, int t, int k) {
    if (k <= 0) return fail(ctx, KMU_E_BAD_K, "kmer_size %d must be positive", k);
    int maxk = 0;
    switch (t) {
    case KMU_KMER32BIT: maxk = 14; break; // src/base/kmer32bit.rs:68-71
    case KMU_KMER16B32BIT: // src/base/kmergenerator.rs:218-220
        if (k != 16) return fail(ctx, KMU_E_BAD_K, "Kmer16b32bit has 16 bases, got %d", k);
        return KMU_OK;
    case KMU_KMER64BIT: maxk = 31; break; // k = 32: push mask is 0 upstream (src/base/kmer64bit.rs:75)
    case KMU_KMERAA32BIT: maxk = 6; break;
    case KMU_KMERAA64BIT: maxk = 12; break;
    default: return fail(ctx, KMU_E_BAD_ARG, "unknown kmer_type %d", t);
    }
    if (k > maxk)
        return fail(ctx, KMU_E_BAD_K, "KmerSeqIterator cannot support so many bases for given kmer type, kmer size %d", k);
    return KMU_OK;
}

bool fhash_valid(int fhash, int t) {
    if (fhash < 0 || fhash > KMU_FHASH_CANON_NTHASH_8B) return false;
    if (kmer_is_aa(t))
        return fhash == KMU_FHASH_IDENTITY_RAW || fhash == KMU_FHASH_VALUE_MASKED || fhash == KMU_FHASH_INVHASH_RAW;
    return true;
}

int stage_sequences(kmu_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, const uint64_t *packed_offsets,
                    uint32_t n_seq, int input_kind, int mem, DevSeqs *out) {
    if (!offsets || (!bases && n_seq)) return fail(ctx, KMU_E_BAD_ARG, "null sequence buffers");
    if (input_kind != KMU_INPUT_ASCII && input_kind != KMU_INPUT_PACKED2)
        return fail(ctx, KMU_E_BAD_ARG, "bad input_kind %d", input_kind);
    if (input_kind == KMU_INPUT_PACKED2 && !packed_offsets)
        return fail(ctx, KMU_E_BAD_ARG, "KMU_INPUT_PACKED2 needs packed_offsets");
    out->n_seq = n_seq;
    out->packed = input_kind == KMU_INPUT_PACKED2;
    if (mem == KMU_MEM_DEVICE) {
        if (((uintptr_t) bases & 15u) != 0) return fail(ctx, KMU_E_BAD_ARG, "device `bases` must be 16-byte aligned");
        out->bases = bases;
        out->offsets = offsets;
        out->packed_offsets = packed_offsets;
        out->total_bytes = 0; // kernels derive it from offsets[n_seq] / packed_offsets
        return KMU_OK;
    }
    if (mem != KMU_MEM_HOST) return fail(ctx, KMU_E_BAD_ARG, "bad mem %d", mem);
    // The sequences may be a range of a larger set (offsets[0] > 0): only their bytes go up, offsets re-based to 0.
    const uint64_t off0 = offsets[0];
    uint64_t first_byte = off0, end_byte = offsets[n_seq];
    if (out->packed) {
        // sequence i occupies packed_offsets[i] .. + ceil(L_i / 4)
        first_byte = n_seq ? packed_offsets[0] : 0;
        end_byte = first_byte;
        for (uint32_t i = 0; i < n_seq; i++) {
            first_byte = std::min(first_byte, packed_offsets[i]);
            end_byte = std::max(end_byte, packed_offsets[i] + (offsets[i + 1] - offsets[i] + 3) / 4);
        }
    }
    const uint64_t total_bytes = end_byte - first_byte;
    const uint64_t *h_off = offsets, *h_poff = packed_offsets;
    if (off0 != 0 || first_byte != 0) {
        out->h_offsets.resize((size_t) n_seq + 1);
        for (uint32_t i = 0; i <= n_seq; i++) out->h_offsets[i] = offsets[i] - off0;
        h_off = out->h_offsets.data();
        if (out->packed) {
            out->h_packed_offsets.resize((size_t) n_seq + 1);
            for (uint32_t i = 0; i < n_seq; i++) out->h_packed_offsets[i] = packed_offsets[i] - first_byte;
            h_poff = out->h_packed_offsets.data();
        }
    }
    void *d_b, *d_o, *d_p = nullptr;
    KMU_TRY(dev_buf(ctx, "in.bases", total_bytes + 64, &d_b));
    KMU_TRY(dev_buf(ctx, "in.offsets", (size_t) (n_seq + 1) * 8, &d_o));
    KMU_HIP(ctx, hipMemcpyAsync(d_b, bases + first_byte, total_bytes, hipMemcpyHostToDevice, ctx->stream));
    KMU_HIP(ctx, hipMemcpyAsync(d_o, h_off, (size_t) (n_seq + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (out->packed) {
        KMU_TRY(dev_buf(ctx, "in.poffsets", (size_t) (n_seq + 1) * 8, &d_p));
        KMU_HIP(ctx, hipMemcpyAsync(d_p, h_poff, (size_t) n_seq * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if (h_off != offsets) KMU_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the temporaries are read by the copies above
    out->bases = (const uint8_t *) d_b;
    out->offsets = (const uint64_t *) d_o;
    out->packed_offsets = (const uint64_t *) d_p;
    out->total_bytes = total_bytes;
    return KMU_OK;
}

// The device error word is STICKY between two reads by the host: a call clears it only when the previous call's bits
// have been read.  In `async_device` contexts KMU_MEM_DEVICE calls return without reading it, so the bits of every call
// since the last read accumulate (kernels only OR into it) and surface at the next call that synchronises --
// kmu_synchronize at the latest.  (Round 1 cleared the word on every entry: a full table or a non-ACGT byte in any call
// but the last of an async sequence was lost.)
int get_err_word(kmu_ctx *ctx, uint32_t **out) {
    void *p;
    const bool fresh = ctx->bufs.find("errword") == ctx->bufs.end() || !ctx->bufs["errword"].p;
    KMU_TRY(dev_buf(ctx, "errword", 64, &p));
    if (fresh || !ctx->err_unread) KMU_HIP(ctx, hipMemsetAsync(p, 0, 64, ctx->stream));
    ctx->err_unread = true;
    *out = (uint32_t *) p;
    return KMU_OK;
}

int check_err_word(kmu_ctx *ctx, uint32_t *d_err) {
    uint32_t h = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&h, d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->err_unread = false; // read: the next call starts from a cleared word
    if (h) KMU_HIP(ctx, hipMemsetAsync(d_err, 0, 4, ctx->stream)); // ... and a second read (kmu_synchronize) does not report it again
    if (h & DERR_NON_ACGT) return fail(ctx, KMU_E_NON_ACGT, "pattern not a code in alphabet_2b (non-ACGT byte in a sequence)");
    if (h & DERR_BAD_AA) return fail(ctx, KMU_E_BAD_ALPHABET, "encode: not a code in alphabet for amino acid");
    if (h & DERR_TABLE_FULL) return fail(ctx, KMU_E_TABLE_FULL, "device hash table full");
    if (h & DERR_BAD_RANGE) return fail(ctx, KMU_E_BAD_ARG, "bad range for kmer iteration (set_range: end <= begin or end > sequence size)");
    if (h & 8u) return fail(ctx, KMU_E_EMPTY_SEQ, "empty sequence (the reference panics in get_nbkmer_guess: ilog2(0))");
    return KMU_OK;
}

int finish_call(kmu_ctx *ctx, int mem) {
    if (mem == KMU_MEM_DEVICE && ctx->async_device) return KMU_OK;
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->profiling) profile_collect(ctx);
    return KMU_OK;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ SeqView make_view(const uint8_t *bases, const uint64_t *offsets,
                                             const uint64_t *packed_offsets, uint32_t n_seq, int packed,
                                             uint64_t total_bytes, uint32_t i) {
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
    return s;
}

// count_non_acgt (src/base/alphabet.rs:28-31): one workgroup per sequence, 16 bytes per lane per step
__global__ void __launch_bounds__(256) k_count_non_acgt(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq,
                                                        uint64_t total, uint64_t *counts) {
    for (uint32_t i = blockIdx.x; i < n_seq; i += gridDim.x) {
        SeqView s = make_view(bases, offsets, nullptr, n_seq, 0, total, i);
        uint64_t nwords = seq_num_words(s);
        uint32_t local = 0;
        for (uint64_t w = threadIdx.x; w < nwords; w += blockDim.x) {
            uint32_t bad;
            (void) load_code_word(s, w, bad);
            local += __popc(bad);
        }
        __shared__ uint32_t acc;
        if (threadIdx.x == 0) acc = 0;
        __syncthreads();
        if (local) atomicAdd(&acc, local);
        __syncthreads();
        if (threadIdx.x == 0) counts[i] = acc;
        __syncthreads();
    }
}

// Sequence::new(raw, 2) (src/base/sequence.rs:48-73): each lane packs 16 bases into 4 output bytes.
// Output byte j of sequence i holds bases 4j..4j+3, first base in bits 7..6, tail padded with 0.
__global__ void __launch_bounds__(256) k_pack2b(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq,
                                                uint64_t total, const uint64_t *packed_offsets, uint8_t *packed,
                                                uint32_t *err) {
    for (uint32_t i = blockIdx.x; i < n_seq; i += gridDim.x) {
        const uint64_t begin = offsets[i], L = offsets[i + 1] - begin;
        const uint64_t nbytes = (L + 3) / 4;
        uint8_t *dst = packed + packed_offsets[i];
        uint32_t anybad = 0;
        for (uint64_t j = threadIdx.x; j < nbytes; j += blockDim.x) {
            uint32_t b = 0;
            for (int t = 0; t < 4; t++) {
                uint64_t p = 4 * j + t;
                uint32_t c = p < L ? bases[begin + p] : (uint32_t) 'A';
                anybad |= !is_acgt(c);
                b |= code2b(c) << (6 - 2 * t);
            }
            dst[j] = (uint8_t) b;
        }
        if (anybad) atomicOr(err, DERR_NON_ACGT);
    }
    (void) total;
}

// per-position k-mer -> fhash(kmer): KmerSeqIterator::next (src/base/kmergenerator.rs:75-106) + closure
// rb / re (both or none): KmerSeqIterator::set_range (kmergenerator.rs:66-68, sequence.rs:562-585) -- only the k-mers that lie
// inside bases [rb[i], re[i]) of sequence i; `Err(())` of the reference (end <= begin or end > L) sets DERR_BAD_RANGE.
__global__ void __launch_bounds__(256) k_kmer_hashes(const uint8_t *bases, const uint64_t *offsets,
                                                     const uint64_t *packed_offsets, uint32_t n_seq, int packed,
                                                     uint64_t total, KmerCfg cfg, const uint64_t *rb, const uint64_t *re,
                                                     uint64_t *out, uint32_t *err) {
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const bool aa = cfg.kmer_type == KMU_KMERAA32BIT || cfg.kmer_type == KMU_KMERAA64BIT;
    for (uint32_t i = blockIdx.x; i < n_seq; i += gridDim.x) {
        SeqView s = make_view(bases, offsets, packed_offsets, n_seq, packed, total, i);
        const uint64_t L = s.len;
        const uint64_t nk = L >= (uint64_t) cfg.k ? L - cfg.k + 1 : 0;
        uint64_t pb = 0, pe = nk; // k-mer start positions [pb, pe)
        if (rb) {
            const uint64_t b = rb[i], e = re[i];
            if (e <= b || e > L) { // sequence.rs:563-565
                if (threadIdx.x == 0) atomicOr(err, DERR_BAD_RANGE);
                continue;
            }
            pb = b;
            pe = e - b >= (uint64_t) cfg.k ? e - cfg.k + 1 : b; // the iterator runs dry once fewer than k bases are left
        }
        uint64_t *o = out + offsets[i];
        uint32_t bad = 0;
        if (pe <= pb) continue;
        if (aa) {
            for (uint64_t st = pb / 64 + wave; st <= (pe - 1) / 64; st += nwaves)
                bad |= wave_step_kmers_aa(s, cfg.k, st, pb, pe, [&](uint64_t p, uint64_t val, uint64_t) {
                    o[p] = apply_fhash(cfg, val, 0);
                });
            if (bad) atomicOr(err, DERR_BAD_AA);
        } else {
            const uint64_t lead = seq_lead(s);
            for (uint64_t st = (pb + lead) / 1024 + wave; st <= (pe - 1 + lead) / 1024; st += nwaves)
                bad |= wave_step_kmers(s, cfg.k, st, pb, pe, [&](uint64_t p, uint64_t val, uint64_t rc) {
                    o[p] = apply_fhash(cfg, val, rc);
                });
            if (bad) atomicOr(err, DERR_NON_ACGT);
        }
    }
}

} // namespace kmu

using namespace kmu;

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *kmu_version(void) { return "kmerutils_amd libkmu 0.1 (gfx950)"; }

int kmu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int kmu_create(const kmu_device_cfg *cfg, kmu_ctx **out) {
    if (!out) return KMU_E_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, KMU_E_NO_DEVICE, "no HIP device: libkmu has no CPU fallback");
    int dev = cfg ? cfg->device_id : 0;
    if (dev < 0 || dev >= n) return fail(nullptr, KMU_E_BAD_ARG, "device_id %d out of range (%d devices)", dev, n);
    if (hipSetDevice(dev) != hipSuccess) return fail(nullptr, KMU_E_HIP, "hipSetDevice(%d) failed", dev);
    kmu_ctx *ctx = new kmu_ctx();
    ctx->device = dev;
    ctx->async_device = cfg && cfg->async_device;
    if (cfg && cfg->stream) {
        ctx->stream = (hipStream_t) cfg->stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return fail(nullptr, KMU_E_HIP, "hipStreamCreate failed");
        }
        ctx->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
        ctx->num_cus = prop.multiProcessorCount;
        ctx->lds_per_block = prop.sharedMemPerBlock;
    }
    *out = ctx;
    return KMU_OK;
}

void kmu_destroy(kmu_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    comm_free(ctx);
    if (ctx->pipe_h2d) (void) hipStreamDestroy(ctx->pipe_h2d);
    if (ctx->pipe_d2h) (void) hipStreamDestroy(ctx->pipe_d2h);
    for (auto &p : ctx->pending) {
        (void) hipEventDestroy(p.a);
        (void) hipEventDestroy(p.b);
    }
    for (auto e : ctx->event_pool) (void) hipEventDestroy(e);
    for (auto &kv : ctx->bufs)
        if (kv.second.p) (void) hipFree(kv.second.p);
    for (auto &kv : ctx->hbufs)
        if (kv.second.p) (void) hipHostFree(kv.second.p);
    if (ctx->own_stream) (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *kmu_last_error(const kmu_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int kmu_synchronize(kmu_ctx *ctx) {
    if (!ctx) return KMU_E_BAD_ARG;
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->profiling) profile_collect(ctx);
    auto it = ctx->bufs.find("errword");
    if (it != ctx->bufs.end() && it->second.p) return check_err_word(ctx, (uint32_t *) it->second.p);
    return KMU_OK;
}

void *kmu_stream(kmu_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }

int kmu_set_hll_params(kmu_ctx *ctx, const kmu_hll_params *hp) {
    if (!ctx || !hp) return KMU_E_BAD_ARG;
    if (!(hp->b > 1.0) || !(hp->a > 0.0) || hp->q == 0 || hp->q >= 0xFFFFFFFEu) return fail(ctx, KMU_E_BAD_ARG, "SetSketchParams: b > 1, a > 0, q >= 1");
    ctx->hll = *hp;
    return KMU_OK;
}

// ---- device buffers for callers without a HIP binding of their own ------------------------------------------------------
int kmu_dev_alloc(kmu_ctx *ctx, uint64_t bytes, void **out) {
    if (!ctx || !out) return KMU_E_BAD_ARG;
    *out = nullptr;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
        (void) hipGetLastError();
        return fail(ctx, KMU_E_OOM, "hipMalloc of %llu bytes failed", (unsigned long long) bytes);
    }
    *out = p;
    return KMU_OK;
}

int kmu_dev_free(kmu_ctx *ctx, void *p) {
    if (!ctx) return KMU_E_BAD_ARG;
    if (!p) return KMU_OK;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream)); // nothing enqueued by this context may still use it
    KMU_HIP(ctx, hipFree(p));
    return KMU_OK;
}

// pinned host memory: what the asynchronous uploads of kmu_sketch_count (and every KMU_MEM_HOST call) want to read from
int kmu_host_alloc(kmu_ctx *ctx, uint64_t bytes, void **out) {
    if (!ctx || !out) return KMU_E_BAD_ARG;
    *out = nullptr;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
        (void) hipGetLastError();
        return fail(ctx, KMU_E_OOM, "hipHostMalloc of %llu bytes failed", (unsigned long long) bytes);
    }
    *out = p;
    return KMU_OK;
}
int kmu_host_free(kmu_ctx *ctx, void *p) {
    if (!ctx) return KMU_E_BAD_ARG;
    if (!p) return KMU_OK;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    KMU_HIP(ctx, hipHostFree(p));
    return KMU_OK;
}

int kmu_copy_to_device(kmu_ctx *ctx, void *dst_device, const void *src_host, uint64_t bytes) {
    if (!ctx || (bytes && (!dst_device || !src_host))) return KMU_E_BAD_ARG;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    KMU_HIP(ctx, hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the host buffer may be reused at return
    return KMU_OK;
}

int kmu_copy_to_host(kmu_ctx *ctx, void *dst_host, const void *src_device, uint64_t bytes) {
    if (!ctx || (bytes && (!dst_host || !src_device))) return KMU_E_BAD_ARG;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    KMU_HIP(ctx, hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KMU_OK;
}

int kmu_profile_enable(kmu_ctx *ctx, int on) {
    if (!ctx) return KMU_E_BAD_ARG;
    if (!on && ctx->profiling) profile_collect(ctx);
    ctx->profiling = on != 0;
    return KMU_OK;
}
int kmu_profile_reset(kmu_ctx *ctx) {
    if (!ctx) return KMU_E_BAD_ARG;
    profile_collect(ctx);
    ctx->stats.clear();
    return KMU_OK;
}
int kmu_profile_get(kmu_ctx *ctx, kmu_kernel_stat *stats, int cap) {
    if (!ctx) return KMU_E_BAD_ARG;
    profile_collect(ctx);
    int n = 0;
    for (auto &kv : ctx->stats) {
        if (stats && n < cap) {
            memset(&stats[n], 0, sizeof(kmu_kernel_stat));
            strncpy(stats[n].name, kv.first.c_str(), sizeof(stats[n].name) - 1);
            stats[n].launches = kv.second.launches;
            stats[n].total_ms = kv.second.ms;
        }
        n++;
    }
    return n;
}

int kmu_count_non_acgt(kmu_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int mem,
                       uint64_t *counts_out) {
    if (!ctx || !counts_out) return KMU_E_BAD_ARG;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, nullptr, n_seq, KMU_INPUT_ASCII, mem, &ds));
    uint64_t *d_counts = counts_out;
    if (mem == KMU_MEM_HOST) {
        void *p;
        KMU_TRY(dev_buf(ctx, "out.u64", (size_t) n_seq * 8 + 8, &p));
        d_counts = (uint64_t *) p;
    }
    if (n_seq) {
        int grid = (int) std::min<uint32_t>(n_seq, (uint32_t) ctx->num_cus * 8);
        KernelTimer t(ctx, "k_count_non_acgt");
        hipLaunchKernelGGL(k_count_non_acgt, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets, n_seq,
                           ds.total_bytes, d_counts);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (mem == KMU_MEM_HOST)
        KMU_HIP(ctx, hipMemcpyAsync(counts_out, d_counts, (size_t) n_seq * 8, hipMemcpyDeviceToHost, ctx->stream));
    return finish_call(ctx, mem);
}

int kmu_pack2b(kmu_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int mem,
               uint8_t *packed_out, uint64_t *packed_offsets_out) {
    if (!ctx || !packed_out || !packed_offsets_out) return KMU_E_BAD_ARG;
    if (mem != KMU_MEM_HOST) return fail(ctx, KMU_E_UNSUPPORTED, "kmu_pack2b: host buffers only (layout needs a host scan)");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    packed_offsets_out[0] = 0;
    for (uint32_t i = 0; i < n_seq; i++)
        packed_offsets_out[i + 1] = packed_offsets_out[i] + (offsets[i + 1] - offsets[i] + 3) / 4;
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, nullptr, n_seq, KMU_INPUT_ASCII, mem, &ds));
    void *d_po, *d_out;
    uint64_t total_out = packed_offsets_out[n_seq];
    KMU_TRY(dev_buf(ctx, "pack.offsets", (size_t) (n_seq + 1) * 8, &d_po));
    KMU_TRY(dev_buf(ctx, "pack.out", total_out + 16, &d_out));
    KMU_HIP(ctx, hipMemcpyAsync(d_po, packed_offsets_out, (size_t) (n_seq + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    if (n_seq) {
        int grid = (int) std::min<uint32_t>(n_seq, (uint32_t) ctx->num_cus * 8);
        KernelTimer t(ctx, "k_pack2b");
        hipLaunchKernelGGL(k_pack2b, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets, n_seq, ds.total_bytes,
                           (const uint64_t *) d_po, (uint8_t *) d_out, d_err);
    }
    KMU_HIP(ctx, hipGetLastError());
    KMU_HIP(ctx, hipMemcpyAsync(packed_out, d_out, total_out, hipMemcpyDeviceToHost, ctx->stream));
    KMU_TRY(check_err_word(ctx, d_err));
    return finish_call(ctx, mem);
}

static int kmer_hashes_impl(kmu_ctx *ctx, const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                            const uint64_t *packed_offsets, uint32_t n_seq, const uint64_t *range_begin,
                            const uint64_t *range_end, uint64_t *out) {
    if (!ctx || !p || !out) return KMU_E_BAD_ARG;
    KMU_TRY(check_kmer(ctx, p->kmer_type, p->kmer_size));
    if (!fhash_valid(p->fhash, p->kmer_type)) return fail(ctx, KMU_E_BAD_ARG, "fhash %d not valid for kmer_type %d", p->fhash, p->kmer_type);
    if (p->input_kind == KMU_INPUT_PACKED2 && (kmer_is_aa(p->kmer_type) || p->fhash == KMU_FHASH_CANON_NTHASH_8B))
        return fail(ctx, KMU_E_BAD_ARG, "packed input not valid for this kmer_type / fhash");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t *d_rb = range_begin, *d_re = range_end;
    if (range_begin && p->mem == KMU_MEM_HOST) {
        for (uint32_t i = 0; i < n_seq; i++) // IterSequence::set_range, sequence.rs:563-565
            if (range_end[i] <= range_begin[i] || range_end[i] > offsets[i + 1] - offsets[i])
                return fail(ctx, KMU_E_BAD_ARG, "bad range [%llu, %llu) for kmer iteration on sequence %u of %llu bases",
                            (unsigned long long) range_begin[i], (unsigned long long) range_end[i], i,
                            (unsigned long long) (offsets[i + 1] - offsets[i]));
        void *q;
        KMU_TRY(dev_buf(ctx, "in.range_b", (size_t) n_seq * 8 + 8, &q));
        KMU_HIP(ctx, hipMemcpyAsync(q, range_begin, (size_t) n_seq * 8, hipMemcpyHostToDevice, ctx->stream));
        d_rb = (const uint64_t *) q;
        KMU_TRY(dev_buf(ctx, "in.range_e", (size_t) n_seq * 8 + 8, &q));
        KMU_HIP(ctx, hipMemcpyAsync(q, range_end, (size_t) n_seq * 8, hipMemcpyHostToDevice, ctx->stream));
        d_re = (const uint64_t *) q;
    }
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, packed_offsets, n_seq, p->input_kind, p->mem, &ds));
    uint64_t *d_out = out;
    uint64_t total = 0;
    const uint64_t off0 = p->mem == KMU_MEM_HOST && n_seq ? offsets[0] : 0; // host ranges are staged re-based to 0
    if (p->mem == KMU_MEM_HOST) {
        total = offsets[n_seq] - off0;
        void *q;
        KMU_TRY(dev_buf(ctx, "out.u64", (size_t) total * 8 + 8, &q));
        d_out = (uint64_t *) q;
        // host arrays come back whole.  Range form: positions outside the ranges keep what the caller's array holds, so it
        // goes up first; plain form: the only positions that start no k-mer are the last k - 1 of every sequence -- they
        // come back as zeros (no upload of 8 bytes per base of possibly uninitialised caller memory for their sake)
        if (range_begin) KMU_HIP(ctx, hipMemcpyAsync(d_out, out + off0, (size_t) total * 8, hipMemcpyHostToDevice, ctx->stream));
        else KMU_HIP(ctx, hipMemsetAsync(d_out, 0, (size_t) total * 8, ctx->stream));
    }
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    if (n_seq) {
        KmerCfg cfg{p->kmer_type, p->kmer_size, p->fhash};
        int grid = (int) std::min<uint32_t>(n_seq, (uint32_t) ctx->num_cus * 8);
        KernelTimer t(ctx, "k_kmer_hashes");
        hipLaunchKernelGGL(k_kmer_hashes, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets, ds.packed_offsets,
                           n_seq, ds.packed, ds.total_bytes, cfg, d_rb, d_re, d_out, d_err);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (p->mem == KMU_MEM_HOST)
        KMU_HIP(ctx, hipMemcpyAsync(out + off0, d_out, (size_t) total * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (!(p->mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
    return finish_call(ctx, p->mem);
}

int kmu_kmer_hashes(kmu_ctx *ctx, const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                    const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *out) {
    return kmer_hashes_impl(ctx, p, bases, offsets, packed_offsets, n_seq, nullptr, nullptr, out);
}

int kmu_kmer_hashes_range(kmu_ctx *ctx, const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                          const uint64_t *packed_offsets, uint32_t n_seq, const uint64_t *range_begin,
                          const uint64_t *range_end, uint64_t *out) {
    if (!range_begin || !range_end) return KMU_E_BAD_ARG;
    return kmer_hashes_impl(ctx, p, bases, offsets, packed_offsets, n_seq, range_begin, range_end, out);
}

int kmu_block_layout(const uint64_t *offsets, uint32_t n_seq, uint32_t block_size, uint64_t *out) {
    if (!offsets || !out || block_size == 0) return KMU_E_BAD_ARG;
    out[0] = 0;
    for (uint32_t i = 0; i < n_seq; i++) {
        uint64_t L = offsets[i + 1] - offsets[i];
        out[i + 1] = out[i] + (L + block_size - 1) / block_size; // src/sketching/seqblocksketch.rs:108-112
    }
    return KMU_OK;
}

} // extern "C"
