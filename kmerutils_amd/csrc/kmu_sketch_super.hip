// kmu_sketch_super.hip -- SuperMinHash / SuperMinHash2 per-sequence kernels for gfx950.
//
// Reference loop (src/sketching/setsketchert.rs:255-297, seqsketchjaccard.rs:328-380): for every k-mer occurrence
// `sminhash.sketch(&fhash(kmer))`: seed an RNG from hasher(value), then for j = 0.. draw r in [0,1) and
// k in [j,m), swap p[j]<->p[k] (Fisher-Yates prefix) and offer r + j to slot p[j]; stop at j > a_upper, the
// largest floor() among the current slot values (Ertl, arXiv 1706.05698 Alg. 3).
// The result is the per-slot minimum over all (item, j) candidates, so items are independent: here every lane
// owns one item at a time and all lanes of the workgroup advance j in lock step; the slot minima live in LDS and
// are updated with ds_min_u64 on the order-preserving bit pattern of the (positive) value; a_upper is refreshed
// from them every few steps (any stale upper bound is valid).  Each lane's partial permutation is an LDS column
// (entry-major, one byte or two per entry) that is restored to the identity after the item through a swap log: the
// swap partners of the first sixteen (eight for two-byte entries) steps of an item sit in four registers -- all active
// lanes are at the same step, so a step's place in the log is wave-uniform -- and an item that needs more steps than
// that (the first items of a short sequence, while slots are still empty) replays its draws.  (Until round 3 the log was
// a second LDS column: with it a wave took 16 KiB at m = 128 and a CU held eight waves.)
#include <algorithm>

#include "kmu_ctx.hpp"
#include "kmu_stream.h"

namespace kmu {

struct SuperArgs {
    const uint8_t *bases;
    const uint64_t *offsets;
    const uint64_t *packed_offsets;
    uint32_t n_seq;
    int packed;
    uint64_t total_bytes;
    KmerCfg cfg;
    int m;
    int lg;       // ceil(log2 m)
    int hasher;
    int rand08;
    int mode;     // 0 f64, 1 f32, 2 u64 (SuperMinHash2), 3 u32 (SuperMinHash2)
    int val_w32;  // Kmer::Val is 32 bits
    uint32_t chunk; // items staged per chunk
    int ncol;       // item lanes that own a permutation column (<= workgroup size)
    const void *hashed;  // pre-hashed input: array of Kmer::Val values (offsets count values, k = 1), else null
    int hashed_bytes;    // 4 / 8
    uint64_t *part_rows; // non-null: raw slot bits per "sequence" (partial results merged by k_super_reduce)
    void *sig_out;
    uint32_t *queue;
    uint32_t *err;
};

__device__ __forceinline__ uint64_t super_init_bits(int mode) {
    switch (mode) {
    case 0: return (uint64_t) __double_as_longlong(4294967295.0); // F::from(u32::MAX)
    case 1: return (uint64_t) __float_as_uint(4294967296.0f);
    case 2: return 0xFFFFFFFFFFFFFFFFull;
    default: return 0xFFFFFFFFull;
    }
}
// min(floor(value), m-1) of a slot bit pattern
__device__ __forceinline__ uint32_t super_floor(uint64_t bits, int mode, int m, int lg) {
    uint32_t f;
    switch (mode) {
    case 0: { double v = __longlong_as_double((long long) bits); f = v >= (double) (m - 1) ? (uint32_t) (m - 1) : (uint32_t) v; break; }
    case 1: { float v = __uint_as_float((uint32_t) bits); f = v >= (float) (m - 1) ? (uint32_t) (m - 1) : (uint32_t) v; break; }
    case 2: { uint64_t v = lg ? bits >> (64 - lg) : 0; f = v >= (uint64_t) (m - 1) ? (uint32_t) (m - 1) : (uint32_t) v; break; }
    default: { uint64_t v = lg ? bits >> (32 - lg) : 0; f = v >= (uint64_t) (m - 1) ? (uint32_t) (m - 1) : (uint32_t) v; break; }
    }
    return f;
}

// MODE (the signature type: 0 f64, 1 f32, 2 u64, 3 u32) and RAND08 (the index draw of rand 0.8) are template constants: the step
// loop below runs ~10 times per item and looked both up in `a` on every step -- two four-way switches and a flag, i.e. chains of
// scalar compares and taken branches around ~110 vector instructions (round 4: config 5's shard 8.4 -> see DESIGN.md 3.3).
template <typename PT, int MODE, bool RAND08>
__global__ void __launch_bounds__(256) k_sketch_super(SuperArgs a) {
#define SUPER_MODE MODE
#define SUPER_R08 RAND08
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int wave = tid >> 6, nwaves = nthreads >> 6;
    const int m = a.m;
    uint64_t *hs = reinterpret_cast<uint64_t *>(smem);          // m slot minima
    uint64_t *items = hs + m;                                     // a.chunk staged RNG seeds
    // step j draws an index in [j, m): the rejection bound of that draw (rand's Uniform: `(0 - range) % range`, or 0.8's zone)
    // depends on j alone -- a table, filled once, instead of an integer division (~20 instructions) in every step
    uint64_t *ztab = items + a.chunk;                             // m bounds
    uint32_t *misc = reinterpret_cast<uint32_t *>(ztab + m);      // [0] read, [1] a_upper
    // one permutation column + swap log per item lane; ncol = nthreads unless the sketch is so large that only some
    // lanes of a single wave get a column
    const int ncol = a.ncol;
    PT *perm = reinterpret_cast<PT *>(misc + 4);                  // [m][ncol]
    constexpr uint32_t LOG_BITS = 8 * sizeof(PT), LOG_PER = 64 / LOG_BITS, LOG_STEPS = 2 * LOG_PER; // swap partners kept in registers
    const bool aa = a.cfg.kmer_type == KMU_KMERAA32BIT || a.cfg.kmer_type == KMU_KMERAA64BIT;
    const uint64_t init_bits = super_init_bits(SUPER_MODE);

    if (tid < ncol)
        for (int e = 0; e < m; e++) perm[(size_t) e * ncol + tid] = (PT) e;
    for (int s = tid; s < m; s += nthreads) hs[s] = init_bits;
    for (int s = tid; s < m; s += nthreads) ztab[s] = Xoshiro::index_bound((uint32_t) (m - s), SUPER_R08);
    if (tid == 0) misc[1] = (uint32_t) (m - 1);
    __syncthreads();

    if (tid == 0) misc[0] = atomicAdd(a.queue, 1u);
    __syncthreads();
    for (;;) {
        const uint32_t r = misc[0];
        __syncthreads(); // everyone holds r before thread 0 posts the next one
        if (r >= a.n_seq) break;
        uint32_t r_next = 0;
        if (tid == 0) r_next = atomicAdd(a.queue, 1u); // its latency hides under this read's work
        SeqView sv;
        sv.base = a.bases;
        sv.len = a.offsets[r + 1] - a.offsets[r];
        sv.packed = a.packed;
        if (a.packed) {
            sv.begin = a.packed_offsets[r];
            sv.total = a.total_bytes ? a.total_bytes
                                     : (a.packed_offsets[a.n_seq - 1] + (a.offsets[a.n_seq] - a.offsets[a.n_seq - 1] + 3) / 4);
        } else {
            sv.begin = a.offsets[r];
            sv.total = a.total_bytes ? a.total_bytes : a.offsets[a.n_seq];
        }
        const uint64_t L = sv.len;
        const uint64_t nk = L >= (uint64_t) a.cfg.k ? L - a.cfg.k + 1 : 0;
        if (L == 0 && tid == 0 && !a.hashed_bytes) atomicOr(a.err, 8u); // an empty list of pre-hashed values is fine
        if (nk == 0 && !a.hashed_bytes && wave_validate_seq(sv, wave, nwaves, aa))
            atomicOr(a.err, aa ? DERR_BAD_AA : DERR_NON_ACGT);
        const uint32_t lead = aa ? 0 : seq_lead(sv);
        for (uint64_t c0 = 0; c0 < nk; c0 += a.chunk) {
            const uint64_t c1 = c0 + a.chunk < nk ? c0 + a.chunk : nk;
            // ---- stage the RNG seeds hasher(fhash(kmer)) of positions [c0, c1) ------------------------------
            uint32_t bad = 0;
            auto visit = [&](uint64_t p, uint64_t val, uint64_t rc) {
                items[p - c0] = hasher_finish(a.hasher, apply_fhash(a.cfg, val, rc), a.val_w32 != 0);
            };
            if (a.hashed_bytes) {
                for (uint64_t p = c0 + tid; p < c1; p += nthreads) {
                    const uint64_t v = a.hashed_bytes == 4 ? (uint64_t) reinterpret_cast<const uint32_t *>(a.hashed)[sv.begin + p]
                                                           : reinterpret_cast<const uint64_t *>(a.hashed)[sv.begin + p];
                    items[p - c0] = hasher_finish(a.hasher, v, a.val_w32 != 0);
                }
            } else if (aa) {
                for (uint64_t st = c0 / 64 + wave; st < (c1 + 63) / 64; st += nwaves)
                    bad |= wave_step_kmers_aa(sv, a.cfg.k, st, c0, c1, visit);
            } else {
                // the last chunk also walks the words that only hold the read's tail, to validate them
                uint64_t st1 = c1 == nk ? (seq_num_words(sv) + 63) / 64 : (c1 - 1 + lead) / 1024 + 1;
                for (uint64_t st = (c0 + lead) / 1024 + wave; st < st1; st += nwaves)
                    bad |= wave_step_kmers(sv, a.cfg.k, st, c0, c1, visit);
            }
            if (bad) atomicOr(a.err, aa ? DERR_BAD_AA : DERR_NON_ACGT);
            __syncthreads();
            // ---- every lane sketches one staged item per batch; j advances in lock step ----------------------
            const uint32_t n_items = (uint32_t) (c1 - c0);
            for (uint32_t b0 = 0; b0 < n_items; b0 += (uint32_t) ncol) {
                const bool have = tid < ncol && b0 + tid < n_items;
                Xoshiro rng;
                if (have) rng.seed(items[b0 + tid]);
                uint32_t j = 0, round = 0;
                uint64_t slog0 = 0, slog1 = 0; // the swap partners of steps 0 .. LOG_STEPS - 1
                bool active = have;
                for (;;) {
                    // (a relaxed atomic load, not a volatile one: the compiler turns a volatile access through a derived pointer into a FLAT load
                    //  with a full vmcnt / lgkmcnt wait -- the step loop had one per step)
                    uint32_t a_upper = __hip_atomic_load(&misc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (active && j > a_upper) active = false;
                    if (!__any(active)) break;
                    if (active) {
                        double rd = 0.0;
                        float rf = 0.f;
                        uint64_t ri = 0;
                        switch (SUPER_MODE) {
                        case 0: rd = rng.unif01(); break;
                        case 1: rf = rng.unif01_f32(); break;
                        case 2: ri = rng.next(); break;
                        default: ri = rng.next_u32(); break;
                        }
                        uint32_t k = rng.unif_index_bound(j, (uint32_t) m, ztab[j], SUPER_R08);
                        const size_t ij = (size_t) j * ncol + tid, ik = (size_t) k * ncol + tid;
                        PT pj = perm[ij];
                        PT pk = perm[ik];
                        perm[ij] = pk;
                        perm[ik] = pj;
                        if (j < LOG_PER) slog0 |= (uint64_t) k << (LOG_BITS * j);
                        else if (j < LOG_STEPS) slog1 |= (uint64_t) k << (LOG_BITS * (j - LOG_PER));
                        uint64_t bits;
                        switch (SUPER_MODE) {
                        case 0: bits = (uint64_t) __double_as_longlong(rd + (double) j); break;
                        case 1: bits = (uint64_t) __float_as_uint(rf + (float) j); break;
                        case 2: bits = a.lg ? (((uint64_t) j << (64 - a.lg)) | (ri >> a.lg)) : ri; break;
                        default: bits = a.lg ? ((((uint64_t) j << (32 - a.lg)) | (ri >> a.lg)) & 0xFFFFFFFFull) : ri; break;
                        }
                        atomicMin((unsigned long long *) &hs[pk], (unsigned long long) bits);
                        j++;
                    }
                    round++; // wave-uniform (all lanes of the wave run this loop together)
                    // (round 4: the waves of a workgroup taking turns at the refresh -- 7.46 against 7.22 ms on config 5's shard: a staler
                    //  bound costs more steps than the refresh saves; every 2 / 8 rounds instead of 4: 7.36 / 7.82 against 7.22)
                    if ((round & 3u) == 0u || round == 1u) { // refresh a_upper = max_s min(floor(hs[s]), m-1)
                        uint32_t mx = 0;
                        for (int s = lane_id(); s < m; s += 64) {
                            uint32_t f = super_floor(__hip_atomic_load(&hs[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), SUPER_MODE, m, a.lg);
                            mx = f > mx ? f : mx;
                        }
                        mx = wave_max_u32(mx);
                        if (lane_id() == 0) atomicMin(&misc[1], mx);
                    }
                }
                // restore this lane's permutation column to the identity
                if (j <= LOG_STEPS) {
                    for (uint32_t jj = 0; jj < j; jj++) {
                        const PT k = (PT) ((jj < LOG_PER ? slog0 >> (LOG_BITS * jj) : slog1 >> (LOG_BITS * (jj - LOG_PER))) & ((1ull << LOG_BITS) - 1ull));
                        perm[(size_t) jj * ncol + tid] = (PT) jj;
                        perm[(size_t) k * ncol + tid] = k;
                    }
                } else { // more steps than the registers log: the draws once more (same generator, same order)
                    Xoshiro r2;
                    r2.seed(items[b0 + tid]);
                    for (uint32_t jj = 0; jj < j; jj++) {
                        switch (SUPER_MODE) {
                        case 0: (void) r2.unif01(); break;
                        case 1: (void) r2.unif01_f32(); break;
                        case 2: (void) r2.next(); break;
                        default: (void) r2.next_u32(); break;
                        }
                        const PT k = (PT) r2.unif_index_bound(jj, (uint32_t) m, ztab[jj], SUPER_R08);
                        perm[(size_t) jj * ncol + tid] = (PT) jj;
                        perm[(size_t) k * ncol + tid] = k;
                    }
                }
            }
            __syncthreads();
        }
        // ---- signature -------------------------------------------------------------------------------------
        for (int t = tid; t < m; t += nthreads) {
            uint64_t v = hs[t];
            if (a.part_rows) {
                a.part_rows[(uint64_t) r * m + t] = v;
                hs[t] = init_bits;
                continue;
            }
            switch (SUPER_MODE) {
            case 0: reinterpret_cast<uint64_t *>(a.sig_out)[(uint64_t) r * m + t] = v; break;
            case 2: reinterpret_cast<uint64_t *>(a.sig_out)[(uint64_t) r * m + t] = v; break;
            default: reinterpret_cast<uint32_t *>(a.sig_out)[(uint64_t) r * m + t] = (uint32_t) v; break;
            }
            hs[t] = init_bits;
        }
        if (tid == 0) { misc[1] = (uint32_t) (m - 1); misc[0] = r_next; }
        __syncthreads();
    }
}
#undef SUPER_MODE
#undef SUPER_R08

// element-wise minimum of the partial slot arrays (order-preserving bit patterns); one thread per slot
// (raw_out: leave the 64-bit slot patterns there instead of a signature)
__global__ void __launch_bounds__(256) k_super_reduce(const uint64_t *part_rows, uint64_t n_parts, int m, int mode,
                                                      void *sig_out, uint64_t *raw_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    uint64_t best = super_init_bits(mode);
    for (uint64_t i = 0; i < n_parts; i++) {
        const uint64_t v = part_rows[i * m + t];
        best = v < best ? v : best;
    }
    if (raw_out) raw_out[t] = best;
    else if (mode == 0 || mode == 2) reinterpret_cast<uint64_t *>(sig_out)[t] = best;
    else reinterpret_cast<uint32_t *>(sig_out)[t] = (uint32_t) best;
}

int launch_super_reduce(kmu_ctx *ctx, const kmu_sketch_params *p, const uint64_t *part_rows, uint64_t n_parts, void *d_sig) {
    int mode = p->algo == KMU_ALGO_SUPER ? (p->sig_type == KMU_SIG_F32 ? 1 : 0) : (p->sig_type == KMU_SIG_U32 ? 3 : 2);
    KernelTimer t(ctx, "k_super_reduce");
    hipLaunchKernelGGL(k_super_reduce, dim3((p->sketch_size + 255) / 256), dim3(256), 0, ctx->stream, part_rows, n_parts,
                       p->sketch_size, mode, d_sig, ctx->partial_out);
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

int launch_super(kmu_ctx *ctx, const kmu_sketch_params *p, const DevSeqs &ds, void *d_sig, uint32_t *d_err,
                 const void *hashed, int hashed_bytes, uint64_t *part_rows) {
    SuperArgs a;
    memset(&a, 0, sizeof a);
    a.hashed = hashed;
    a.hashed_bytes = hashed_bytes;
    a.part_rows = part_rows;
    a.bases = ds.bases;
    a.offsets = ds.offsets;
    a.packed_offsets = ds.packed_offsets;
    a.n_seq = ds.n_seq;
    a.packed = ds.packed;
    a.total_bytes = ds.total_bytes;
    a.cfg = KmerCfg{p->kmer_type, hashed_bytes ? 1 : p->kmer_size, p->fhash};
    a.m = p->sketch_size;
    a.lg = 0;
    while ((1 << a.lg) < a.m) a.lg++;
    a.hasher = p->hasher;
    a.rand08 = (p->flags & KMU_FLAG_RAND08) ? 1 : 0;
    if (p->algo == KMU_ALGO_SUPER) a.mode = p->sig_type == KMU_SIG_F32 ? 1 : 0;
    else a.mode = p->sig_type == KMU_SIG_U32 ? 3 : 2;
    a.val_w32 = kmer_val_bytes(p->kmer_type) == 4;
    a.sig_out = d_sig;
    a.err = d_err;
    void *q;
    KMU_TRY(dev_buf(ctx, "queue", 64, &q));
    KMU_HIP(ctx, hipMemsetAsync(q, 0, 64, ctx->stream));
    a.queue = (uint32_t *) q;
    // Round 1 chose one wave per workgroup and a staging chunk of 256 items (config 5: 27.6 -> 12.8 ms, scripts/dbg_super.sh: the
    // step loop then waited on FLAT loads and divisions).  Round 4: the items of a batch advance j in lock step against ONE bound
    // (a_upper falls with every offer of every lane), so the more items of a read are in flight together the fewer rounds the
    // read takes -- four waves and 512 staged items: 7.93 -> 7.25 ms on config 5's shard (scripts/r04_superthreads.sh; 1 024
    // staged items: 8.7).
    a.chunk = 512;
    const bool wide = a.m > 256;
    const size_t pt = wide ? 2 : 1;
    const size_t lds_max = 160 * 1024;
    typedef void (*super_kernel_t)(SuperArgs);
    super_kernel_t kern = nullptr;
#define KMU_SUPER_PICK(PT_) \
    switch (a.mode * 2 + a.rand08) { \
    case 0: kern = k_sketch_super<PT_, 0, false>; break; case 1: kern = k_sketch_super<PT_, 0, true>; break; \
    case 2: kern = k_sketch_super<PT_, 1, false>; break; case 3: kern = k_sketch_super<PT_, 1, true>; break; \
    case 4: kern = k_sketch_super<PT_, 2, false>; break; case 5: kern = k_sketch_super<PT_, 2, true>; break; \
    case 6: kern = k_sketch_super<PT_, 3, false>; break; default: kern = k_sketch_super<PT_, 3, true>; break; \
    }
    if (wide) { KMU_SUPER_PICK(uint16_t) } else { KMU_SUPER_PICK(uint8_t) }
#undef KMU_SUPER_PICK
    auto fn = (const void *) kern;
    // the per-lane permutation columns dominate the LDS footprint: shrink the workgroup for large m
    int threads = 256;
    size_t lds = 0;
    int ncol = 0;
    for (; threads >= 64; threads -= 64) {
        lds = (size_t) 16 * a.m + (size_t) 8 * a.chunk + 16 + pt * a.m * threads;
        lds = (lds + 15) & ~(size_t) 15;
        if (lds <= lds_max) { ncol = threads; break; }
    }
    if (!ncol) { // very large sketches: one wave, as many item lanes as columns fit
        threads = 64;
        const size_t fixed = (size_t) 16 * a.m + (size_t) 8 * a.chunk + 16 + 16;
        if (fixed < lds_max) ncol = (int) std::min<size_t>(64, (lds_max - fixed) / (pt * (size_t) a.m));
        lds = (fixed + pt * (size_t) a.m * ncol + 15) & ~(size_t) 15;
    }
    a.ncol = ncol;
    if (ncol < 1) return fail(ctx, KMU_E_UNSUPPORTED, "sketch_size %d too large for LDS (%zu B)", a.m, lds);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_max) != hipSuccess) {
            (void) hipGetLastError();
            return fail(ctx, KMU_E_UNSUPPORTED, "sketch_size %d needs %zu B of LDS", a.m, lds);
        }
    }
    int blocks_per_cu = std::max<int>(1, std::min<int>(2048 / threads, (int) (lds_max / lds)));
    int grid = (int) std::min<uint64_t>((uint64_t) ds.n_seq, (uint64_t) ctx->num_cus * blocks_per_cu);
    if (grid < 1) grid = 1;
    {
        KernelTimer t(ctx, "k_sketch_super");
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, ctx->stream, a);
    }
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

} // namespace kmu
