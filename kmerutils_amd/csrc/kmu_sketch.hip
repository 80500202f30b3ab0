// kmu_sketch.hip -- per-sequence sketching kernels (ProbMinHash3a / SuperMinHash / bottom-k) for gfx950.
//
// Reference loop being replaced (src/sketching/seqsketchjaccard.rs:224-243, setsketchert.rs:121-157):
//     for every read (rayon):  FnvHashMap<Val,u64> of fhash(kmer) over all k-mers  ->  ProbMinHash3a(m)
// MI355X mapping: one persistent workgroup per CU pulls reads from an atomic queue.  The read's weighted
// multiset lives in an LDS open-addressing table (8-byte key + 4-byte count per slot, ~13k slots in 160 KiB);
// reads with more distinct k-mers than the table holds are processed in P hash-partitions (each pass rescans
// the read and keeps the keys of its partition; counts stay exact because a key always lands in one partition).
// The m slot minima (h as order-preserving f64 bits, arg-min key) stay in LDS across passes.
// Integer / f64 ALU + LDS atomics only; HBM traffic = the read's bases in, m signatures out.
#include <algorithm>
#include <cmath>

#include "kmu_ctx.hpp"
#include "kmu_stream.h"

namespace kmu {

static constexpr uint64_t KEY_EMPTY = 0xFFFFFFFFFFFFFFFFull; // LDS table sentinel (the one real key equal to it is
                                                             // counted in a side word)
static constexpr uint64_t H_INIT = 0x7FEFFFFFFFFFFFFFull;    // bits of f64::MAX (MaxValueTracker initial value)
static constexpr uint64_t H_BUSY = 0xFFFFFFFFFFFFFFFEull;    // slot being updated

struct SketchArgs {
    const uint8_t *bases;
    const uint64_t *offsets;
    const uint64_t *packed_offsets;
    const uint64_t *block_rows; // block mode: row offsets per read, else null
    uint32_t n_seq;
    int packed;
    uint64_t total_bytes;
    KmerCfg cfg;
    int m;
    int hasher;
    int rand08;
    int sig_bytes;   // 4 or 8
    uint32_t block_size;
    uint32_t table_slots; // S
    uint32_t part_cap;    // max k-mers handled by one pass
    Exp01 e01;
    void *sig_out;
    uint32_t *queue; // atomic read counter
    uint32_t *err;
};

__device__ __forceinline__ uint32_t mix32(uint64_t key) {
    uint32_t x = (uint32_t) key ^ (uint32_t) (key >> 32);
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    return x;
}
__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t) (((uint64_t) a * b) >> 32); }

// LDS multiset insert: keys[] (u64, KEY_EMPTY = free) + cnt[] (u32).  Returns false when the table is full.
__device__ __forceinline__ bool table_insert(uint64_t *keys, uint32_t *cnt, uint32_t S, uint64_t key, uint32_t h) {
    uint32_t s = mulhi32(h, S);
    for (uint32_t probes = 0; probes < S; probes++) {
        unsigned long long old = atomicCAS((unsigned long long *) &keys[s], (unsigned long long) KEY_EMPTY,
                                           (unsigned long long) key);
        if (old == KEY_EMPTY || old == key) {
            atomicAdd(&cnt[s], 1u);
            return true;
        }
        s = s + 1 == S ? 0 : s + 1;
    }
    return false;
}

// slot update: keep (h, key) minimal per slot; exact ties go to the smaller key (order independence)
__device__ __forceinline__ void slot_update(uint64_t *hmin, uint64_t *sig, uint32_t k, double h, uint64_t key) {
    const uint64_t hb = (uint64_t) __double_as_longlong(h);
    for (;;) {
        uint64_t cur = *(volatile uint64_t *) &hmin[k];
        if (cur == H_BUSY) continue;
        if (hb > cur) return;
        if (hb == cur && key >= *(volatile uint64_t *) &sig[k]) return;
        if (atomicCAS((unsigned long long *) &hmin[k], (unsigned long long) cur, (unsigned long long) H_BUSY) == cur) {
            *(volatile uint64_t *) &sig[k] = key;
            __threadfence_block();
            atomicExch((unsigned long long *) &hmin[k], (unsigned long long) hb);
            return;
        }
    }
}

// q_max = max over slots of the current minima (MaxValueTracker root); a slot in flight counts as "unknown" = MAX
__device__ __forceinline__ uint64_t wave_qmax(const uint64_t *hmin, int m) {
    uint64_t q = 0;
    for (int i = lane_id(); i < m; i += 64) {
        uint64_t v = *(volatile const uint64_t *) &hmin[i];
        if (v == H_BUSY) v = H_INIT;
        q = v > q ? v : q;
    }
    return wave_max_u64(q);
}

// ProbMinHash3a for one key of weight w: every generated point (h, slot) is offered to the slot minima.
// `qmax_bits` is any upper bound of the current q_max; pruning with it never changes the arg-min.
__device__ __forceinline__ void pmh3a_consume_wave(const SketchArgs &a, uint64_t *hmin, uint64_t *sig, bool have,
                                                   uint64_t key, uint32_t w) {
    Xoshiro rng;
    double winv = 0.0;
    bool alive = have;
    if (have) {
        rng.seed(hasher_finish(KMU_HASHER_NOHASH, key, a.sig_bytes == 4));
        winv = 1.0 / (double) w;
    }
    uint32_t i = 1;
    uint64_t qb = wave_qmax(hmin, a.m);
    while (__any(alive)) {
        if (alive) {
            double qmax = __longlong_as_double((long long) qb);
            double hbase = winv * (double) (i - 1);
            if (!(hbase < qmax)) {
                alive = false;
            } else {
                double x = exp01_sample(a.e01, rng);
                double h = hbase + winv * x;
                uint32_t k = rng.unif_index(0, (uint32_t) a.m, a.rand08 != 0);
                if (h < qmax) slot_update(hmin, sig, k, h, key);
                else if (i == 1) alive = false; // first point already beyond q_max: the reference drops the key
                if (!(winv * (double) i < qmax)) alive = false;
                i++;
            }
        }
        if (__any(alive)) qb = wave_qmax(hmin, a.m);
    }
}

// One workgroup = one read at a time (or one block of a read in block mode).
__global__ void __launch_bounds__(1024) k_sketch_pmh3a(SketchArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t S = a.table_slots;
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    uint64_t *hmin = keys + S;
    uint64_t *sig = hmin + a.m;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(sig + a.m);
    uint32_t *misc = cnt + S; // [0] current read, [1] count of KEY_EMPTY-valued keys, [2] flags
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int wave = tid >> 6, nwaves = nthreads >> 6;
    const bool aa = a.cfg.kmer_type == KMU_KMERAA32BIT || a.cfg.kmer_type == KMU_KMERAA64BIT;

    for (uint32_t s = tid; s < S; s += nthreads) { keys[s] = KEY_EMPTY; cnt[s] = 0; }
    for (int s = tid; s < a.m; s += nthreads) { hmin[s] = H_INIT; sig[s] = 0; }
    if (tid == 0) { misc[1] = 0; misc[2] = 0; }
    __syncthreads();

    for (;;) {
        if (tid == 0) misc[0] = atomicAdd(a.queue, 1u);
        __syncthreads();
        const uint32_t r = misc[0];
        if (r >= a.n_seq) break;
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
        const uint64_t nk_all = L >= (uint64_t) a.cfg.k ? L - a.cfg.k + 1 : 0;
        if (L == 0 && tid == 0) atomicOr(a.err, 8u);
        if (nk_all == 0 && wave_validate_seq(sv, wave, nwaves, aa)) atomicOr(a.err, aa ? DERR_BAD_AA : DERR_NON_ACGT);
        const uint64_t nsteps = aa ? (L + 63) / 64 : (seq_num_words(sv) + 63) / 64;
        // blocks of the read (src/sketching/seqblocksketch.rs:108-146); whole read = one block
        const uint64_t B = a.block_size ? a.block_size : (nk_all ? nk_all : 1);
        const uint64_t nblocks = a.block_size ? (L + B - 1) / B : 1;
        for (uint64_t blk = 0; blk < nblocks; blk++) {
            uint64_t pb = blk * B, pe = pb + B;
            if (pb > nk_all) pb = nk_all;
            if (pe > nk_all) pe = nk_all;
            const uint64_t nk = pe - pb;
            const uint32_t P = nk ? (uint32_t) ((nk + a.part_cap - 1) / a.part_cap) : 0;
            // table region used by this block: small reads only touch (and later sweep) a prefix of the table
            uint32_t Seff = S;
            if (P == 1) {
                uint64_t want = ((nk * 8) / 5 + 64 + 63) & ~63ull;
                Seff = want < S ? (uint32_t) want : S;
            }
            // only the steps that hold positions [pb, pe + k - 1) matter
            uint64_t st0 = 0, st1 = nsteps;
            if (a.block_size && !aa) {
                uint32_t lead = seq_lead(sv);
                st0 = (pb + lead) / 1024;
                st1 = nk ? ((pe - 1 + lead) / 1024 + 1) : st0;
            } else if (a.block_size) {
                st0 = pb / 64;
                st1 = nk ? ((pe - 1) / 64 + 1) : st0;
            }
            for (uint32_t part = 0; part < P; part++) {
                // ---- pass A: multiset of the keys of this partition -------------------------------------
                uint32_t bad = 0;
                bool full = false;
                auto visit = [&](uint64_t, uint64_t val, uint64_t rc) {
                    uint64_t key = apply_fhash(a.cfg, val, rc);
                    uint32_t h = mix32(key);
                    if (P > 1 && mulhi32(h * 0x85EBCA6Bu, P) != part) return;
                    if (key == KEY_EMPTY) { atomicAdd(&misc[1], 1u); return; }
                    if (!table_insert(keys, cnt, Seff, key, h)) full = true;
                };
                for (uint64_t st = st0 + wave; st < st1; st += nwaves) {
                    if (aa) bad |= wave_step_kmers_aa(sv, a.cfg.k, st, pb, pe, visit);
                    else bad |= wave_step_kmers(sv, a.cfg.k, st, pb, pe, visit);
                }
                if (bad) atomicOr(a.err, aa ? DERR_BAD_AA : DERR_NON_ACGT);
                if (full) atomicOr(a.err, DERR_TABLE_FULL);
                __syncthreads();
                // ---- pass B: every distinct key generates its points; table is swept clean ---------------
                for (uint32_t base = wave * 64; base < Seff; base += nwaves * 64) {
                    uint32_t s = base + lane_id();
                    uint64_t key = keys[s];
                    uint32_t w = cnt[s];
                    bool have = key != KEY_EMPTY;
                    if (have) { keys[s] = KEY_EMPTY; cnt[s] = 0; }
                    if (__any(have)) pmh3a_consume_wave(a, hmin, sig, have, key, w);
                }
                if (wave == 0 && misc[1] != 0) { // the key whose value equals the sentinel
                    uint32_t w = misc[1];
                    pmh3a_consume_wave(a, hmin, sig, lane_id() == 0, KEY_EMPTY, w);
                }
                __syncthreads();
                if (tid == 0) misc[1] = 0;
            }
            // ---- signature of this block: arg-min key per slot, initobj (0) for an empty multiset -----------
            {
                uint64_t row = a.block_rows ? a.block_rows[r] + blk : (uint64_t) r;
                for (int t = tid; t < a.m; t += nthreads) {
                    uint64_t v = hmin[t] == H_INIT ? 0ull : sig[t];
                    if (a.sig_bytes == 4) reinterpret_cast<uint32_t *>(a.sig_out)[row * a.m + t] = (uint32_t) v;
                    else reinterpret_cast<uint64_t *>(a.sig_out)[row * a.m + t] = v;
                    hmin[t] = H_INIT;
                    sig[t] = 0;
                }
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

} // namespace kmu

using namespace kmu;

static int sketch_params_check(kmu_ctx *ctx, const kmu_sketch_params *p) {
    if (p->sketch_size < (p->algo == KMU_ALGO_BOTTOMK ? 1 : 2) || p->sketch_size > 65536)
        return fail(ctx, KMU_E_BAD_ARG, "sketch_size %d out of range", p->sketch_size);
    int w = kmer_val_bytes(p->kmer_type);
    switch (p->algo) {
    case KMU_ALGO_PROB3A:
        if (p->sig_type != (w == 4 ? KMU_SIG_U32 : KMU_SIG_U64))
            return fail(ctx, KMU_E_BAD_ARG, "ProbMinHash3a signature type is Kmer::Val (setsketchert.rs:107)");
        if (p->hasher != KMU_HASHER_NOHASH)
            return fail(ctx, KMU_E_BAD_ARG, "ProbMinHash3a is always built with NoHashHasher (seqsketchjaccard.rs:235)");
        break;
    case KMU_ALGO_SUPER:
        if (p->sig_type != KMU_SIG_F32 && p->sig_type != KMU_SIG_F64) return fail(ctx, KMU_E_BAD_ARG, "SuperMinHash signature is f32/f64");
        break;
    case KMU_ALGO_SUPER2:
        if (p->sig_type != KMU_SIG_U32 && p->sig_type != KMU_SIG_U64) return fail(ctx, KMU_E_BAD_ARG, "SuperMinHash2 signature is u32/u64");
        break;
    case KMU_ALGO_BOTTOMK:
        if (p->sig_type != KMU_SIG_U64) return fail(ctx, KMU_E_BAD_ARG, "bottom-k rows are u64 hashes");
        break;
    default: return fail(ctx, KMU_E_BAD_ARG, "unknown algo %d", p->algo);
    }
    if (p->block_size < 0) return fail(ctx, KMU_E_BAD_ARG, "negative block_size");
    if (p->block_size > 0 && (p->algo != KMU_ALGO_PROB3A || p->mode != KMU_MODE_PER_SEQ))
        return fail(ctx, KMU_E_UNSUPPORTED, "block sketching is ProbMinHash3a per sequence (seqblocksketch.rs:97)");
    return KMU_OK;
}

namespace kmu {
int launch_super(kmu_ctx *ctx, const kmu_sketch_params *p, const DevSeqs &ds, void *d_sig, uint32_t *d_err);
int launch_bottomk(kmu_ctx *ctx, const kmu_sketch_params *p, const DevSeqs &ds, void *d_sig, uint32_t *d_counts,
                   uint32_t *d_err);
}

static int launch_pmh3a(kmu_ctx *ctx, const kmu_sketch_params *p, const DevSeqs &ds, const uint64_t *d_block_rows,
                        void *d_sig, uint32_t *d_err) {
    SketchArgs a;
    memset(&a, 0, sizeof a);
    a.bases = ds.bases;
    a.offsets = ds.offsets;
    a.packed_offsets = ds.packed_offsets;
    a.block_rows = d_block_rows;
    a.n_seq = ds.n_seq;
    a.packed = ds.packed;
    a.total_bytes = ds.total_bytes;
    a.cfg = KmerCfg{p->kmer_type, p->kmer_size, p->fhash};
    a.m = p->sketch_size;
    a.hasher = p->hasher;
    a.rand08 = (p->flags & KMU_FLAG_RAND08) ? 1 : 0;
    a.sig_bytes = kmer_val_bytes(p->kmer_type);
    a.block_size = (uint32_t) p->block_size;
    // ExpRestricted01::new(lambda), lambda = ln(m / (m-1)) -- same libm expressions as the crate / the oracle
    double lambda = std::log((double) a.m / (double) (a.m - 1));
    a.e01.lambda = lambda;
    a.e01.c1 = (std::exp(lambda) - 1.0) / lambda;
    a.e01.c2 = std::log(2.0 / (1.0 + std::exp(-lambda))) / lambda;
    a.e01.c3 = (1.0 - std::exp(-lambda)) / lambda;
    a.sig_out = d_sig;
    a.err = d_err;
    // LDS budget: keys 8S + cnt 4S + 16 m + 64
    size_t lds_max = 160 * 1024;
    if (hipFuncSetAttribute((const void *) k_sketch_pmh3a, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_max) !=
        hipSuccess) {
        (void) hipGetLastError();
        lds_max = 64 * 1024;
    }
    size_t fixed = (size_t) 16 * a.m + 64;
    if (fixed + 12 * 256 > lds_max) return fail(ctx, KMU_E_UNSUPPORTED, "sketch_size %d too large for LDS", a.m);
    uint32_t S = (uint32_t) (((lds_max - fixed) / 12) & ~63ull);
    const char *env = getenv("KMU_PMH_SLOTS");
    if (env && atoi(env) >= 256) S = std::min<uint32_t>(S, (uint32_t) atoi(env) & ~63u);
    a.table_slots = S;
    a.part_cap = (uint32_t) ((uint64_t) S * 5 / 8);
    size_t lds = (size_t) 12 * S + fixed;
    void *q;
    KMU_TRY(dev_buf(ctx, "queue", 64, &q));
    KMU_HIP(ctx, hipMemsetAsync(q, 0, 64, ctx->stream));
    a.queue = (uint32_t *) q;
    int threads = 1024;
    const char *tenv = getenv("KMU_PMH_THREADS");
    if (tenv && atoi(tenv) >= 64) threads = std::min(1024, atoi(tenv) & ~63);
    int blocks_per_cu = std::max<int>(1, (int) (lds_max / lds));
    int grid = (int) std::min<uint64_t>((uint64_t) ds.n_seq, (uint64_t) ctx->num_cus * blocks_per_cu);
    if (grid < 1) grid = 1;
    {
        KernelTimer t(ctx, "k_sketch_pmh3a");
        hipLaunchKernelGGL(k_sketch_pmh3a, dim3(grid), dim3(threads), lds, ctx->stream, a);
    }
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

extern "C" int kmu_sketch(kmu_ctx *ctx, const kmu_sketch_params *p, const uint8_t *bases, const uint64_t *offsets,
                          const uint64_t *packed_offsets, uint32_t n_seq, const uint64_t *block_row_offsets,
                          void *sig_out, uint32_t *counts_out) {
    if (!ctx || !p || !sig_out) return KMU_E_BAD_ARG;
    KMU_TRY(check_kmer(ctx, p->kmer_type, p->kmer_size));
    KMU_TRY(sketch_params_check(ctx, p));
    if (!fhash_valid(p->fhash, p->kmer_type)) return fail(ctx, KMU_E_BAD_ARG, "fhash %d not valid for kmer_type %d", p->fhash, p->kmer_type);
    if (p->input_kind == KMU_INPUT_PACKED2 && (kmer_is_aa(p->kmer_type) || p->fhash == KMU_FHASH_CANON_NTHASH_8B))
        return fail(ctx, KMU_E_BAD_ARG, "packed input not valid for this kmer_type / fhash");
    if (p->mode == KMU_MODE_ALL_SEQS)
        return fail(ctx, KMU_E_UNSUPPORTED, "sketch_compressedkmer_seqs (one sketch for all sequences) is not built yet");
    if (p->block_size > 0 && !block_row_offsets) return fail(ctx, KMU_E_BAD_ARG, "block mode needs block_row_offsets");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, packed_offsets, n_seq, p->input_kind, p->mem, &ds));
    const size_t sigb = (p->sig_type == KMU_SIG_U32 || p->sig_type == KMU_SIG_F32) ? 4 : 8;
    uint64_t rows = n_seq;
    const uint64_t *d_block_rows = nullptr;
    void *d_sig = sig_out;
    uint32_t *d_counts = counts_out;
    if (p->mem == KMU_MEM_HOST) {
        if (p->block_size > 0) {
            rows = block_row_offsets[n_seq];
            void *q;
            KMU_TRY(dev_buf(ctx, "in.blockrows", (size_t) (n_seq + 1) * 8, &q));
            KMU_HIP(ctx, hipMemcpyAsync(q, block_row_offsets, (size_t) (n_seq + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
            d_block_rows = (const uint64_t *) q;
        }
        void *q;
        KMU_TRY(dev_buf(ctx, "out.sig", rows * p->sketch_size * sigb + 64, &q));
        d_sig = q;
        if (counts_out) {
            KMU_TRY(dev_buf(ctx, "out.counts", rows * p->sketch_size * 4 + 64, &q));
            d_counts = (uint32_t *) q;
        }
    } else {
        d_block_rows = block_row_offsets;
    }
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    if (n_seq) {
        switch (p->algo) {
        case KMU_ALGO_PROB3A: KMU_TRY(launch_pmh3a(ctx, p, ds, d_block_rows, d_sig, d_err)); break;
        case KMU_ALGO_SUPER:
        case KMU_ALGO_SUPER2: KMU_TRY(launch_super(ctx, p, ds, d_sig, d_err)); break;
        case KMU_ALGO_BOTTOMK: KMU_TRY(launch_bottomk(ctx, p, ds, d_sig, d_counts, d_err)); break;
        }
    }
    if (p->mem == KMU_MEM_HOST) {
        KMU_HIP(ctx, hipMemcpyAsync(sig_out, d_sig, rows * p->sketch_size * sigb, hipMemcpyDeviceToHost, ctx->stream));
        if (counts_out)
            KMU_HIP(ctx, hipMemcpyAsync(counts_out, d_counts, rows * p->sketch_size * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!(p->mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
    return finish_call(ctx, p->mem);
}

extern "C" int kmu_sketch_hashed(kmu_ctx *ctx, const kmu_sketch_params *p, const void *hashed, const uint64_t *offsets,
                                 uint32_t n_seq, void *sig_out, uint32_t *counts_out) {
    (void) hashed; (void) offsets; (void) n_seq; (void) sig_out; (void) counts_out; (void) p;
    if (!ctx) return KMU_E_BAD_ARG;
    return fail(ctx, KMU_E_UNSUPPORTED, "kmu_sketch_hashed is not built yet");
}
