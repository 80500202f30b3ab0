// kmu_sketch.hip -- the host side of the per-sequence sketch path: which kernels a batch takes (routes chosen per call from the
// batch's shape), their launches and scratch buffers, the chunked host pipeline of kmu_sketch_count, and the C entry points
// kmu_sketch / kmu_sketch_hashed / kmu_sketch_count / kmu_sketch_partial / kmu_sketch_merge_partials / kmu_kmer_hashes_compact.
// Reference: SeqSketcherT::sketch_compressedkmer / sketch_compressedkmer_seqs (src/sketching/setsketchert.rs:54-80,
// seqsketchjaccard.rs:211-319), datasketcher's loop (src/bin/datasketcher.rs:222-226).  The kernels are in kmu_sketch_kernels.hip.
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "kmu_hostpack.hpp"
#include "kmu_sketch_kernels.h"

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
    case KMU_ALGO_OPTDENS:
    case KMU_ALGO_REVOPTDENS:
        if (p->sig_type != KMU_SIG_F32 && p->sig_type != KMU_SIG_F64) return fail(ctx, KMU_E_BAD_ARG, "OptDens / RevOptDens signature is f32/f64");
        break;
    case KMU_ALGO_HLL:
        if (p->sig_type != KMU_SIG_U16 && p->sig_type != KMU_SIG_U32 && p->sig_type != KMU_SIG_U64)
            return fail(ctx, KMU_E_BAD_ARG, "SetSketch registers are u16 / u32 / u64");
        if (p->sig_type == KMU_SIG_U16 && ctx->hll.q + 1 > 65535u) return fail(ctx, KMU_E_BAD_ARG, "q + 1 does not fit u16 registers");
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
int launch_super(kmu_ctx *ctx, const kmu_sketch_params *p, const DevSeqs &ds, void *d_sig, uint32_t *d_err,
                 const void *hashed, int hashed_bytes, uint64_t *part_rows);
int launch_super_reduce(kmu_ctx *ctx, const kmu_sketch_params *p, const uint64_t *part_rows, uint64_t n_parts, void *d_sig);
int launch_dens(kmu_ctx *ctx, const kmu_sketch_params *p, const DevSeqs &ds, void *d_sig, uint32_t *d_err, const void *hashed,
                int hashed_bytes);
int launch_dens_merge(kmu_ctx *ctx, const kmu_sketch_params *p, const uint64_t *parts, uint32_t n_parts, void *d_sig);
}
// the sketches kept as m bins / registers with one independent update per k-mer occurrence (kmu_sketch_dens.hip)
static bool algo_is_dens(int algo) { return algo == KMU_ALGO_OPTDENS || algo == KMU_ALGO_REVOPTDENS || algo == KMU_ALGO_HLL; }

// ProbMinHash3a of whole DNA sequences with k <= 8 (Kmer32bit) and a closure that is injective on the (canonical) k-mer:
// the histogram route of k_sketch_smallk.  KMU_PMH_SMALLK=0 keeps the general kernels (diagnostics, A/B).
static bool smallk_route(const kmu_sketch_params *p, int hashed_bytes, bool partial, bool blocks) {
    if (p->algo != KMU_ALGO_PROB3A || p->kmer_type != KMU_KMER32BIT || p->kmer_size > 8 || hashed_bytes || partial || blocks ||
        p->block_size != 0 || p->sketch_size > 512)
        return false;
    switch (p->fhash) {
    case KMU_FHASH_IDENTITY_RAW: case KMU_FHASH_VALUE_MASKED: case KMU_FHASH_CANON_RAW: case KMU_FHASH_CANON_INVHASH:
    case KMU_FHASH_INVHASH_RAW: case KMU_FHASH_CANON_VALUE: break;
    default: return false; // (ntHash is not injective in principle)
    }
    const char *e = getenv("KMU_PMH_SMALLK");
    return !(e && atoi(e) == 0);
}

// k_pmh_points over the lists of a.n_seq reads; reads with more than KMU_PMH_PTS_LONG list entries (default 32 768; 0: none)
// are listed first (k_pts_long_list) and taken by whole workgroups
static int launch_points(kmu_ctx *ctx, SketchArgs a, int cus) {
    void (*const kpts)(SketchArgs) = a.sig_bytes == 4 ? k_pmh_points<true> : k_pmh_points<false>;
    const size_t lds2 = (size_t) 4 * (2 * (size_t) a.m + PTS_WAVE_WORDS) * 8 + WINV_LUT * 8;
    if (lds2 > 64 * 1024)
        KMU_HIP(ctx, hipFuncSetAttribute((const void *) kpts, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int per_cu = (int) std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds2));
    const int grid2 = (int) std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t) a.n_seq + 3) / 4, (uint64_t) cus * per_cu)); // (cus: see KMU_PMH_RESERVE_CUS)
    uint32_t thr = 32768; // (bench: the device leg is the same with or without; the host leg's chunks gain 1.3 ms of 128; 16 384: +0.4 ms on the device leg, 8 192: +2)
    if (const char *e = getenv("KMU_PMH_PTS_LONG")) thr = (uint32_t) std::max(0, atoi(e));
    a.pts_long = nullptr;
    a.pts_long_t = thr;
    if (thr && a.n_seq) {
        if (thr < 1024u) a.pts_long_t = thr = 1024u; // (a workgroup's four waves all need chunks of their own)
        void *pl;
        KMU_TRY(dev_buf(ctx, "pts.long", ((size_t) a.n_seq + 2) * 4 + 64, &pl));
        KMU_HIP(ctx, hipMemsetAsync(pl, 0, 8, ctx->stream));
        a.pts_long = (uint32_t *) pl;
        hipLaunchKernelGGL(k_pts_long_list, dim3((a.n_seq + 255) / 256), dim3(256), 0, ctx->stream, (const uint32_t *) a.lst_n, a.n_seq, thr,
                           (uint32_t *) pl);
    }
    KernelTimer t(ctx, "k_pmh_points");
    hipLaunchKernelGGL(kpts, dim3(grid2), dim3(256), lds2, ctx->stream, a);
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

static int launch_pmh3a(kmu_ctx *ctx, const kmu_sketch_params *p, const DevSeqs &ds, const uint64_t *d_block_rows,
                        void *d_sig, uint32_t *d_counts, uint32_t *d_err, const void *hashed = nullptr,
                        int hashed_bytes = 0, uint64_t *part_h = nullptr, uint64_t *part_k = nullptr,
                        uint32_t skip_longer = 0, const uint64_t *len_stats = nullptr /* longest sequence, all bases */) {
    const bool bottomk = p->algo == KMU_ALGO_BOTTOMK;
    SketchArgs a;
    memset(&a, 0, sizeof a);
    a.skip_longer = skip_longer;
    a.hashed = hashed;
    a.hashed_bytes = hashed_bytes;
    a.part_h = part_h;
    a.part_k = part_k;
    a.bases = ds.bases;
    a.offsets = ds.offsets;
    a.packed_offsets = ds.packed_offsets;
    a.block_rows = d_block_rows;
    a.n_seq = ds.n_seq;
    a.n_queue = ds.n_seq;
    a.packed = ds.packed;
    a.total_bytes = ds.total_bytes;
    a.cfg = KmerCfg{p->kmer_type, hashed_bytes ? 1 : p->kmer_size, p->fhash};
    a.m = p->sketch_size;
    a.hasher = p->hasher;
    a.rand08 = (p->flags & KMU_FLAG_RAND08) ? 1 : 0;
    a.sig_bytes = kmer_val_bytes(p->kmer_type);
    a.block_size = (uint32_t) p->block_size;
    {
        uint32_t m32 = (uint32_t) a.m;
        a.idx_thresh = (0u - m32) % m32;
        uint64_t m64 = (uint64_t) a.m;
        a.idx_zone = 0xFFFFFFFFFFFFFFFFull - (0xFFFFFFFFFFFFFFFFull - m64 + 1ull) % m64;
    }
    // ExpRestricted01::new(lambda), lambda = ln(m / (m-1)) -- same libm expressions as the crate / the oracle
    double lambda = a.m >= 2 ? std::log((double) a.m / (double) (a.m - 1)) : 1.0;
    a.e01.lambda = lambda;
    a.e01.c1 = (std::exp(lambda) - 1.0) / lambda;
    a.e01.c2 = std::log(2.0 / (1.0 + std::exp(-lambda))) / lambda;
    a.e01.c3 = (1.0 - std::exp(-lambda)) / lambda;
    a.sig_out = d_sig;
    a.err = d_err;
    a.ablate = 0u;
#if KMU_DIAG // (diagnostic builds: parts of the kernels switched off / clocked, scripts/r05_pts_parts.sh)
    if (const char *ab = getenv("KMU_PMH_ABLATE")) a.ablate = (uint32_t) atoi(ab);
#endif
    const bool aa = kmer_is_aa(p->kmer_type) || hashed_bytes != 0; // pre-hashed values use the byte-stream instantiation
    typedef void (*sketch_kernel_t)(SketchArgs);
    if (smallk_route(p, hashed_bytes, part_h != nullptr, d_block_rows != nullptr)) { // k <= 8: direct-indexed histogram
        // the distinct (key, weight) pairs go to k_pmh_points through lists in HBM (12 bytes per base of scratch, the lists
        // of the two-kernel route) unless that memory is not to be had: then the histogram kernel makes the points itself
        bool emit = (size_t) 4 * (2 * (size_t) p->sketch_size + PTS_WAVE_WORDS) * 8 + WINV_LUT * 8 <= 150 * 1024;
        if (const char *e = getenv("KMU_PMH_SPLIT")) emit = emit && atoi(e) != 0;
        uint64_t total = 0;
        if (emit) {
            if (len_stats) total = len_stats[1];
            else if (!ds.h_offsets.empty()) total = ds.h_offsets[ds.n_seq] - ds.h_offsets[0];
            else {
                uint64_t ends[2] = {0, 0};
                KMU_HIP(ctx, hipMemcpyAsync(&ends[0], ds.offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
                KMU_HIP(ctx, hipMemcpyAsync(&ends[1], ds.offsets + ds.n_seq, 8, hipMemcpyDeviceToHost, ctx->stream));
                KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
                total = ends[1] - ends[0];
            }
            const size_t need_k = total * 8 + 64, need_w = total * 4 + 64;
            size_t grow = 0, free_b = 0, total_b = 0;
            if (ctx->bufs["cnt.partA"].bytes < need_k) grow += need_k + need_k / 8;
            if (ctx->bufs["pmh.lst_w"].bytes < need_w) grow += need_w + need_w / 8;
            if (grow && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < grow + total_b / 8) emit = false;
            if (emit) {
                void *lk, *lw, *ln;
                KMU_TRY(dev_buf(ctx, "cnt.partA", need_k, &lk));
                KMU_TRY(dev_buf(ctx, "pmh.lst_w", need_w, &lw));
                KMU_TRY(dev_buf(ctx, "pmh.lst_n", (size_t) ds.n_seq * 8 + 64, &ln));
                a.lst_keys = (uint64_t *) lk;
                a.lst_w = (uint32_t *) lw;
                a.lst_n = (uint32_t *) ln;
                a.lst_nu = a.lst_n + ds.n_seq;
                KMU_HIP(ctx, hipMemsetAsync(a.lst_nu, 0, (size_t) ds.n_seq * 4, ctx->stream));
            }
        }
        // (Round 4, measured and not kept: the first point of every one of the 4^k possible keys from a table made once per call -- no
        //  generator in pass 1 -- but 4e9 gathers of 16 bytes out of a 1 MB table are 4e9 lines from L2: k_pmh_points 26.1 against 18.9 ms
        //  on config 3; 16-bit lower bounds of the samples in LDS in front of the gather: 52 ms.)
        const sketch_kernel_t kern = emit ? k_sketch_smallk<true> : k_sketch_smallk<false>;
        // LDS: histogram | slot minima (only when the kernel makes the points itself) | list of u16 indices | staged words
        const size_t lds_fixed = (size_t) SMALLK_WORDS * 4 + (emit ? 0 : (size_t) 16 * a.m) + ((size_t) SMALLK_TILE + 2) * 4 + 64;
        a.cap = (uint32_t) ((160 * 1024 - lds_fixed) / 2) & ~2047u;
        if (a.cap > 32768u) a.cap = 32768u;
        const size_t lds = lds_fixed + (size_t) a.cap * 2;
        KMU_HIP(ctx, hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        void *q;
        KMU_TRY(dev_buf(ctx, "queue", 256, &q));
        KMU_HIP(ctx, hipMemsetAsync(q, 0, 256, ctx->stream));
        a.queue = (uint32_t *) q;
        a.queue2 = a.queue + 48;
        int cus = ctx->num_cus;
        if (const char *rs = getenv("KMU_PMH_RESERVE_CUS")) cus = std::max(cus / 2, cus - std::max(0, atoi(rs)));
        const int grid = (int) std::max<uint64_t>(1, std::min<uint64_t>((uint64_t) ds.n_seq, (uint64_t) cus));
        {
            KernelTimer t(ctx, "k_sketch_smallk");
            hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, ctx->stream, a);
        }
        KMU_HIP(ctx, hipGetLastError());
        if (emit) {
            KMU_TRY(launch_points(ctx, a, cus));
        }
        return KMU_OK;
    }
    // Big batches of whole DNA sequences go through two kernels: the multiset kernel leaves the (key, weight) pairs of
    // every read in HBM, k_pmh_points (one wave per read, no workgroup barrier, 5 waves per SIMD) generates the points.
    // ONT workload: 53.3 + 21.2 ms against 88.7 ms in one kernel, for 12 bytes of scratch per base.  One wave per read
    // has a tail: the longest read keeps its wave busy while the others have run out of reads.  The route is taken
    // when the gain (16 % of the single kernel's time) exceeds the expected overhang of that read; figures of an MI355X
    // (a wave of k_pmh_points does 4.0e4 k-mers per ms, the single kernel 4.9e7 per ms with 256 CUs).
    // KMU_PMH_SPLIT = 0 / 1: never / always.
    const char *split_env = getenv("KMU_PMH_SPLIT");
    const int split_mode = split_env ? atoi(split_env) : -1;
    bool split = !bottomk && !aa && !part_h && !d_block_rows && p->block_size == 0 && !skip_longer &&
                 (size_t) 4 * (2 * (size_t) p->sketch_size + PTS_WAVE_WORDS) * 8 + WINV_LUT * 8 <= 150 * 1024 && // four waves' arrays fit one workgroup
                 split_mode != 0 && (split_mode == 1 || len_stats);
    uint64_t list_total = 0; // number of bases = capacity of the (key, weight) lists
    if (split) {
        uint64_t &total = list_total;
        if (len_stats) total = len_stats[1];
        else if (!ds.h_offsets.empty()) total = ds.h_offsets[ds.n_seq] - ds.h_offsets[0];
        else {
            uint64_t ends[2] = {0, 0};
            KMU_HIP(ctx, hipMemcpyAsync(&ends[0], ds.offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
            KMU_HIP(ctx, hipMemcpyAsync(&ends[1], ds.offsets + ds.n_seq, 8, hipMemcpyDeviceToHost, ctx->stream));
            KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
            total = ends[1] - ends[0];
        }
        if (split_mode != 1) {
            const double cu_share = (double) ctx->num_cus / 256.0;
            // (a wave of k_pmh_points does 4.0e4 k-mers per ms; a read beyond KMU_PMH_PTS_LONG is taken by four)
            const char *ple = getenv("KMU_PMH_PTS_LONG");
            const uint64_t pts_thr = ple ? (uint64_t) std::max(0, atoi(ple)) : 32768u;
            const double wave_rate = pts_thr && len_stats[0] > pts_thr ? 1.6e5 : 4.0e4;
            const double t_ideal = (double) total / (2.35e8 * cu_share), t_tail = (double) len_stats[0] / wave_rate; // ms
            const double overhang = t_tail >= t_ideal ? t_tail - 0.5 * t_ideal : t_tail * t_tail / (2.0 * t_ideal);
            // (r02: with the reads that fit a workgroup's registers on k_multiset_uq the two-kernel route takes 53 ms where the
            //  single kernel takes 87 on the ONT workload: 39 % of the single kernel's time, 16 % before)
            // (r03: 49.8 ms, 43 %; the points kernel 18.5 ms for 4.36 G k-mers)
            const double gain = 0.43 * (double) total / (4.9e7 * cu_share);
            if (gain <= overhang + 0.02) split = false; // (0.02 ms: the second launch)
        }
    }
    if (split) {
        const uint64_t total = list_total;
        // 8 bytes per base: the same scratch the count build uses for its first partition level ("cnt.partA"); a
        // context never runs the two at the same time, and at 4.4 Gbases per GPU a second copy would not fit next to
        // the count table and the exchange buffers
        const size_t need_k = total * 8 + 64, need_w = total * 4 + 64;
        size_t grow = 0;
        if (ctx->bufs["cnt.partA"].bytes < need_k) grow += need_k + need_k / 8;
        if (ctx->bufs["pmh.lst_w"].bytes < need_w) grow += need_w + need_w / 8;
        size_t free_b = 0, total_b = 0;
        if (grow && split_mode != 1 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < grow + total_b / 8)
            split = false; // the lists would crowd out what comes after this call: one kernel, no lists
        if (split) {
            void *lk, *lw, *ln;
            KMU_TRY(dev_buf(ctx, "cnt.partA", need_k, &lk));
            KMU_TRY(dev_buf(ctx, "pmh.lst_w", need_w, &lw));
            KMU_TRY(dev_buf(ctx, "pmh.lst_n", (size_t) ds.n_seq * 8 + 64, &ln));
            a.lst_keys = (uint64_t *) lk;
            a.lst_w = (uint32_t *) lw;
            a.lst_n = (uint32_t *) ln;
            a.lst_nu = a.lst_n + ds.n_seq;
            KMU_HIP(ctx, hipMemsetAsync(a.lst_nu, 0, (size_t) ds.n_seq * 4, ctx->stream));
        }
    }
    const char *plain_env = getenv("KMU_PMH_PLAIN"); // diagnostics: 0 = always the general instantiation
    const bool plain = !bottomk && !aa && !part_h && !d_block_rows && p->block_size == 0 && !ds.packed &&
                 !(plain_env && atoi(plain_env) == 0);
    const sketch_kernel_t kern = bottomk ? (aa ? k_sketch_pmh3a<true, true> : k_sketch_pmh3a<false, true>)
                                 : aa    ? k_sketch_pmh3a<true, false>
                                 : split ? (plain ? k_sketch_pmh3a<false, false, true, true> : k_sketch_pmh3a<false, false, true>)
                                 : plain ? k_sketch_pmh3a<false, false, false, true>
                                         : k_sketch_pmh3a<false, false>;
    const void *fn = (const void *) kern;
    a.counts_out = d_counts;
    a.bk_shift = (a.sig_bytes == 4 && p->hasher == KMU_HASHER_NOHASH) ? 20 : 52; // NoHashHasher of a u32 is < 2^32
    a.bk_mask = p->hasher == KMU_HASHER_INT64HASH ? 0xFFu : 0xFFFFu;
    size_t lds_max = 160 * 1024;
    {
        hipFuncAttributes fa;
        KMU_HIP(ctx, hipFuncGetAttributes(&fa, fn));
        lds_max -= fa.sharedSizeBytes; // static LDS (none today) comes out of the same 160 KiB
    }
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_max) != hipSuccess) {
        (void) hipGetLastError();
        lds_max = 64 * 1024;
    }
    const sketch_kernel_t kern_redo = k_sketch_pmh3a<false, false>; // takes what the PLAIN instantiation hands back
    if (plain && lds_max > 64 * 1024 &&
        hipFuncSetAttribute((const void *) kern_redo, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_max) != hipSuccess)
        return fail(ctx, KMU_E_HIP, "hipFuncSetAttribute failed for the general sketch kernel");
    // LDS budget: dense keys 8 cap | weights 4 cap | slot minima 16 m | buckets 4 (NB+1) | misc | staged words
    // (bottom-k re-uses the staged-word area for its per-bucket distinct counts: NBUCKETS + 1 words)
    a.tile_words = (aa && !bottomk) ? 4 : (lds_max > 64 * 1024 || bottomk ? 4096 + 2 : 1024 + 2);
    size_t fixed = (size_t) 16 * a.m + 4 * ((size_t) NBUCKETS + 1 + 8) + 4 * (M_WORDS + 16 + DEF_PARTS) +
                   4 * ((size_t) a.tile_words + 4) + 64;
    if (fixed + 12 * 256 > lds_max) return fail(ctx, KMU_E_UNSUPPORTED, "sketch_size %d too large for LDS", a.m);
    uint32_t cap = (uint32_t) ((lds_max - fixed) / 12);
    cap &= ~63u;
    if (cap > 65472) cap = 65472; // positions are stored in 16 bits
    a.cap = cap;
    a.part_target = cap - cap / 10;
    a.inv_part_target = 1.0 / (double) a.part_target;
    if (plain && a.part_target > (uint32_t) KREG * 1024u)
        return fail(ctx, KMU_E_HIP, "internal: a single pass (%u k-mers) must fit the register keys of the PLAIN kernel", a.part_target);
    {
        const uint32_t tp = (a.tile_words - 2) * 16; // 4096 or 1024 words of 16 bases
        a.tile_shift = 0;
        while ((1u << a.tile_shift) < tp) a.tile_shift++;
        if ((1u << a.tile_shift) != tp) return fail(ctx, KMU_E_HIP, "internal: tile size %u is not a power of two", tp);
    }
    size_t lds = (size_t) 12 * cap + fixed;
    void *q;
    KMU_TRY(dev_buf(ctx, "queue", 256, &q)); // [0] read cursor; u64 words 8..23: phase clocks (diagnostics)
    KMU_HIP(ctx, hipMemsetAsync(q, 0, 256, ctx->stream));
    a.queue = (uint32_t *) q;
    const int threads = 1024;
    int blocks_per_cu = std::max<int>(1, (int) (lds_max / lds));
    int cus = ctx->num_cus;
    if (const char *rs = getenv("KMU_PMH_RESERVE_CUS")) // CUs left to concurrent work (RCCL kernels of an exchange in flight)
        cus = std::max(cus / 2, cus - std::max(0, atoi(rs)));
    int grid = (int) std::min<uint64_t>((uint64_t) ds.n_seq, (uint64_t) cus * blocks_per_cu);
    if (grid < 1) grid = 1;
    {
        void *sk, *si, *sw, *bk = nullptr, *bc = nullptr;
        KMU_TRY(dev_buf(ctx, "pmh.scr_keys", (size_t) grid * cap * 8, &sk));
        KMU_TRY(dev_buf(ctx, "pmh.scr_info", (size_t) grid * cap * 4, &si));
        KMU_TRY(dev_buf(ctx, "pmh.scr_w", (size_t) grid * cap * 4, &sw));
        a.scr_keys = (uint64_t *) sk;
        a.scr_info = (uint32_t *) si;
        a.scr_w = (uint32_t *) sw;
        void *dkq;
        KMU_TRY(dev_buf(ctx, "pmh.def_keys", (size_t) grid * DEF_CAP * 8, &dkq));
        a.def_keys = (uint64_t *) dkq;
        a.bk_keys = (uint64_t *) bk;
        a.bk_cnt = (uint32_t *) bc;
    }
    if (split) a.queue2 = a.queue + 48;
    if (plain) {
        void *rl;
        KMU_TRY(dev_buf(ctx, "pmh.redo", (size_t) ds.n_seq * 4 + 64, &rl));
        a.redo_list = (uint32_t *) rl;
    }
    // Whole unpacked DNA reads on the two-kernel route: the reads that fit one workgroup's registers (<= 10 240 k-mers: 87 % of
    // the reads, 64 % of the bases of the ONT workload) go through k_multiset_uq, which does not sort what occurs once; the
    // longer ones (and the rare read with too many repeated keys) are handed to the general list-emitting kernel.
    const char *uq_env = getenv("KMU_PMH_UQ"); // 0: every read through the counting-sort kernel (A/B)
    const bool uq = split && plain && !(uq_env && atoi(uq_env) == 0);
    bool main_launched = false, short_route = false;
    const char *sh_env = getenv("KMU_PMH_SHORT"); // 0: short reads through k_multiset_uq like the others (A/B)
    if (uq && len_stats && len_stats[0] < (uint64_t) SHORT_KEYS + (uint64_t) p->kmer_size && !(sh_env && atoi(sh_env) == 0)) {
        // every read of the batch has at most 256 k-mers: one wave per read (k_multiset_short)
        const size_t lds_s = 4 * SHORT_WAVE_BYTES;
        const int per_cu = 5; // (88 registers: five waves per SIMD)
        KernelTimer t(ctx, "k_multiset_short");
        hipLaunchKernelGGL(k_multiset_short, dim3((unsigned) std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t) ds.n_seq + 3) / 4, (uint64_t) cus * per_cu))),
                           dim3(256), lds_s, ctx->stream, a);
        KMU_HIP(ctx, hipGetLastError());
        KMU_HIP(ctx, hipMemsetAsync(a.queue, 0, 256, ctx->stream)); // (the points kernel's cursor)
        main_launched = true;
        short_route = true;
    } else if (uq) {
        {
            const auto ka = k_multiset_uq<512, UQ1_BM, UQ1_COLL, 4>;
            const size_t lds_a = UqShape<512, UQ1_BM, UQ1_COLL>::LDS;
            KMU_HIP(ctx, hipFuncSetAttribute((const void *) ka, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            KernelTimer t(ctx, "k_multiset_uq");
            hipLaunchKernelGGL(ka, dim3((unsigned) std::max<uint64_t>(1, std::min<uint64_t>(ds.n_seq, (uint64_t) cus * 2))), dim3(512), lds_a,
                               ctx->stream, a);
        }
        KMU_HIP(ctx, hipGetLastError());
        uint32_t n_long = 0;
        KMU_HIP(ctx, hipMemcpyAsync(&n_long, a.queue + 56, 4, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ABL(256u)) { // diagnostics: thread-0 clocks of the first shape, share per phase
            unsigned long long ph[10];
            KMU_HIP(ctx, hipMemcpy(ph, (const uint64_t *) a.queue + 8, sizeof ph, hipMemcpyDeviceToHost));
            static const char *nm[9] = {"stage+wipe", "B1", "keys+bitmaps", "B2", "next fetch", "sort out", "B3", "collisions", "end"};
            double tot = 0;
            for (int i = 0; i < 9; i++) tot += (double) ph[i];
            fprintf(stderr, "[kmu uq phases]");
            for (int i = 0; i < 9; i++) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * (double) ph[i] / (tot > 0 ? tot : 1));
            fprintf(stderr, "  (%.3g clocks)\n", tot);
        }
        if (n_long) { // the second shape: reads of up to 20 480 k-mers, from the first one's list
            const auto kb = k_multiset_uq<1024, UQ2_BM, UQ2_COLL, 4>;
            const size_t lds_b = UqShape<1024, UQ2_BM, UQ2_COLL>::LDS;
            void *rl2;
            KMU_TRY(dev_buf(ctx, "pmh.redo2", (size_t) ds.n_seq * 4 + 64, &rl2));
            KMU_HIP(ctx, hipFuncSetAttribute((const void *) kb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            KMU_HIP(ctx, hipMemsetAsync(a.queue, 0, 256, ctx->stream));
            uint32_t *list1 = a.redo_list;
            a.read_list = list1;
            a.redo_list = (uint32_t *) rl2;
            a.n_queue = n_long;
            {
                KernelTimer t(ctx, "k_multiset_uq");
                hipLaunchKernelGGL(kb, dim3((unsigned) std::max<uint64_t>(1, std::min<uint64_t>(n_long, (uint64_t) cus))), dim3(1024), lds_b, ctx->stream, a);
            }
            KMU_HIP(ctx, hipGetLastError());
            KMU_HIP(ctx, hipMemcpyAsync(&n_long, a.queue + 56, 4, hipMemcpyDeviceToHost, ctx->stream));
            KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
            a.read_list = nullptr;
            a.n_queue = ds.n_seq;
        }
        if (n_long) {
            const sketch_kernel_t kgen = k_sketch_pmh3a<false, false, true>; // reads its reads from a list; repetitive reads in rounds
            if (lds_max > 64 * 1024 && hipFuncSetAttribute((const void *) kgen, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_max) != hipSuccess)
                return fail(ctx, KMU_E_HIP, "hipFuncSetAttribute failed for the general list-emitting kernel");
            KMU_HIP(ctx, hipMemsetAsync(a.queue, 0, 256, ctx->stream));
            a.read_list = a.redo_list;
            a.n_queue = n_long;
            const int gridl = (int) std::min<uint64_t>((uint64_t) n_long, (uint64_t) cus * blocks_per_cu);
            KernelTimer t(ctx, "k_sketch_pmh3a");
            hipLaunchKernelGGL(kgen, dim3(gridl), dim3(threads), lds, ctx->stream, a);
            KMU_HIP(ctx, hipGetLastError());
            a.read_list = nullptr;
            a.n_queue = ds.n_seq;
            if (ABL(256u)) { // diagnostics: the phases of the list-emitting launch (the words are wiped for the points kernel below)
                unsigned long long ph[10];
                KMU_HIP(ctx, hipMemcpy(ph, (const uint64_t *) a.queue + 8, sizeof ph, hipMemcpyDeviceToHost));
                static const char *nm[10] = {"header", "stage", "A1", "scan", "place", "A3", "B1", "B2+clear", "row", "next+parked"};
                double tot = 0;
                for (int i = 0; i < 10; i++) tot += (double) ph[i];
                fprintf(stderr, "[kmu list phases] grid %d, %u reads:", gridl, n_long);
                for (int i = 0; i < 10; i++) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * (double) ph[i] / (tot > 0 ? tot : 1));
                fprintf(stderr, "  (clocks/wg %.3g)\n", tot / gridl);
            }
        }
        KMU_HIP(ctx, hipMemsetAsync(a.queue, 0, 256, ctx->stream)); // (the points kernel's cursor; nothing is left to redo)
        main_launched = true;
    }
    if (!main_launched) {
        KernelTimer t(ctx, bottomk ? "k_sketch_bottomk" : "k_sketch_pmh3a");
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, ctx->stream, a);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (split && short_route) { // lists of at most 256 pairs: all keys of a read in a wave's registers, round by round
        const sketch_kernel_t kpts = a.sig_bytes == 4 ? k_pmh_points_short<true> : k_pmh_points_short<false>;
        const size_t lds2 = (size_t) 4 * (2 * (size_t) a.m + 2) * 8 + WINV_LUT * 8;
        if (lds2 > 64 * 1024)
            KMU_HIP(ctx, hipFuncSetAttribute((const void *) kpts, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        const int per_cu = (int) std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / lds2));
        const int grid2 = (int) std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t) ds.n_seq + 3) / 4, (uint64_t) cus * per_cu));
        KernelTimer t(ctx, "k_pmh_points_short");
        hipLaunchKernelGGL(kpts, dim3(grid2), dim3(256), lds2, ctx->stream, a);
        KMU_HIP(ctx, hipGetLastError());
    } else if (split) {
        KMU_TRY(launch_points(ctx, a, cus));
    }
    if (plain && !uq) { // (after the points kernel, whose row for such a sequence is empty)
        // sequences whose k-mers overflowed a pass (repetitive ones): the general instantiation redoes them in rounds
        uint32_t n_redo = 0;
        KMU_HIP(ctx, hipMemcpyAsync(&n_redo, a.queue + 56, 4, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (n_redo) {
            KMU_HIP(ctx, hipMemsetAsync(a.queue, 0, 256, ctx->stream));
            a.read_list = a.redo_list;
            a.n_queue = n_redo;
            const int grid2 = (int) std::min<uint64_t>((uint64_t) n_redo, (uint64_t) cus * blocks_per_cu);
            KernelTimer t(ctx, "k_sketch_pmh3a_redo");
            hipLaunchKernelGGL(kern_redo, dim3(grid2), dim3(threads), lds, ctx->stream, a);
            KMU_HIP(ctx, hipGetLastError());
        }
    }
    if (ABL(256u)) { // diagnostics: mean clocks per workgroup and phase
        unsigned long long ph[10];
        KMU_HIP(ctx, hipMemcpyAsync(ph, (const uint64_t *) a.queue + 8, sizeof ph, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        static const char *nm[10] = {"header", "stage", "A1", "scan", "place", "A3", "B1", "B2+clear", "row", "next+parked"};
        double tot = 0;
        for (int i = 0; i < 10; i++) tot += (double) ph[i];
        fprintf(stderr, "[kmu phases] grid %d:", grid);
        for (int i = 0; i < 10; i++) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * (double) ph[i] / (tot > 0 ? tot : 1));
        fprintf(stderr, "  (clocks/wg %.3g)\n", tot / grid);
    }
    return KMU_OK;
}

// One sketch over a device array of n pre-hashed values (u64, zero-extended Kmer::Val).
//  ProbMinHash3a: the keys are radix-partitioned by hash into leaves that fit the LDS multiset; every leaf is an
//  independent weighted set (disjoint keys), sketched into partial slot minima, merged per slot by (h, key).
//  SuperMinHash(2): items are independent; chunks are sketched separately and the slot values merged by min.
static int sketch_all_hashed(kmu_ctx *ctx, const kmu_sketch_params *p, const uint64_t *d_vals, uint64_t n, void *d_sig,
                             uint32_t *d_err) {
    const int m = p->sketch_size;
    if (p->algo == KMU_ALGO_PROB3A) {
        // leaves of ~4k keys: comfortably inside one LDS pass even with a skewed hash
        int region_bits = 0;
        while (region_bits < 22 && (n >> region_bits) > 4096) region_bits++;
        const uint64_t *items, *bounds;
        KMU_TRY(partition_u64(ctx, d_vals, n, region_bits, &items, &bounds));
        const uint64_t n_leaves = 1ull << region_bits;
        void *ph, *pk;
        KMU_TRY(dev_buf(ctx, "all.part_h", n_leaves * m * 8, &ph));
        KMU_TRY(dev_buf(ctx, "all.part_k", n_leaves * m * 8, &pk));
        DevSeqs leaves;
        leaves.bases = reinterpret_cast<const uint8_t *>(items);
        leaves.offsets = bounds;
        leaves.n_seq = (uint32_t) n_leaves;
        leaves.total_bytes = 1; // unused for pre-hashed input
        KMU_TRY(launch_pmh3a(ctx, p, leaves, nullptr, nullptr, nullptr, d_err, items, 8, (uint64_t *) ph, (uint64_t *) pk));
        {
            KernelTimer t(ctx, "k_pmh_reduce");
            hipLaunchKernelGGL(k_pmh_reduce, dim3(m), dim3(256), 0, ctx->stream, (const uint64_t *) ph, (const uint64_t *) pk,
                               n_leaves, m, (uint64_t) m, kmer_val_bytes(p->kmer_type), d_sig, ctx->partial_out);
        }
        KMU_HIP(ctx, hipGetLastError());
        return KMU_OK;
    }
    // SuperMinHash / SuperMinHash2
    const uint64_t per = 16384;
    const uint64_t n_chunks = std::max<uint64_t>(1, std::min<uint64_t>((n + per - 1) / per, 8192));
    std::vector<uint64_t> h_off(n_chunks + 1);
    for (uint64_t c = 0; c <= n_chunks; c++) h_off[c] = n * c / n_chunks;
    void *d_off, *pr;
    KMU_TRY(dev_buf(ctx, "all.chunk_off", (n_chunks + 1) * 8, &d_off));
    KMU_HIP(ctx, hipMemcpyAsync(d_off, h_off.data(), (n_chunks + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream)); // h_off is a local
    KMU_TRY(dev_buf(ctx, "all.part_rows", n_chunks * m * 8, &pr));
    DevSeqs chunks;
    chunks.bases = reinterpret_cast<const uint8_t *>(d_vals);
    chunks.offsets = (const uint64_t *) d_off;
    chunks.n_seq = (uint32_t) n_chunks;
    chunks.total_bytes = 1;
    KMU_TRY(launch_super(ctx, p, chunks, nullptr, d_err, d_vals, 8, (uint64_t *) pr));
    return launch_super_reduce(ctx, p, (const uint64_t *) pr, n_chunks, d_sig);
}

// ProbMinHash3a, one signature per sequence, sequences on the device.  h_offsets: a host copy of ds.offsets[0 .. n_seq] if the
// caller has one (the lengths are then known without asking the device), else null.
static int sketch_pmh_per_seq(kmu_ctx *ctx, const kmu_sketch_params *p, const DevSeqs &ds, const uint64_t *d_block_rows, void *d_sig,
                              uint32_t *d_err, const uint64_t *h_offsets) {
    const uint32_t n_seq = ds.n_seq;
    const size_t sigb = p->sig_type == KMU_SIG_U16 ? 2 : (p->sig_type == KMU_SIG_U32 || p->sig_type == KMU_SIG_F32) ? 4 : 8;
    // Sequences far longer than one LDS pass (genomes, not reads) would take L / cap passes in the per-sequence
    // kernel.  They go through the same global route as a sketch over all sequences -- hashes, radix partition
    // into leaves, per-leaf slot minima, merge -- one sequence at a time, which is linear in L.
    uint32_t skip_longer = 0;
    uint64_t len_stats[2] = {0, 0}; // longest sequence, all bases (whole sequences only)
    std::vector<uint32_t> long_seqs;
    if (p->block_size == 0 && smallk_route(p, 0, false, d_block_rows != nullptr)) // any length fits the histogram
        return launch_pmh3a(ctx, p, ds, d_block_rows, d_sig, nullptr, d_err);
    if (p->block_size == 0) {
        std::vector<uint64_t> h_off;
        if (h_offsets) {
            for (uint32_t i = 0; i < n_seq; i++) len_stats[0] = std::max(len_stats[0], h_offsets[i + 1] - h_offsets[i]);
            len_stats[1] = h_offsets[n_seq] - h_offsets[0];
        } else {
            void *mx;
            KMU_TRY(dev_buf(ctx, "pmh.maxlen", 64, &mx));
            KMU_HIP(ctx, hipMemsetAsync(mx, 0, 8, ctx->stream));
            const uint32_t mgrid = (uint32_t) std::min<uint64_t>(((uint64_t) n_seq + 1023) / 1024, (uint64_t) ctx->num_cus);
            hipLaunchKernelGGL(k_max_len, dim3(mgrid ? mgrid : 1), dim3(1024), 0, ctx->stream, ds.offsets, n_seq, (uint64_t *) mx);
            KMU_HIP(ctx, hipMemcpyAsync(len_stats, mx, 16, hipMemcpyDeviceToHost, ctx->stream));
            KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        const uint64_t max_len = len_stats[0];
        if (max_len > (uint64_t) LONG_SEQ_KMERS + (uint64_t) p->kmer_size) {
            if (!h_offsets) {
                h_off.resize((size_t) n_seq + 1);
                KMU_HIP(ctx, hipMemcpyAsync(h_off.data(), ds.offsets, ((size_t) n_seq + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
                KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
                h_offsets = h_off.data();
            }
            for (uint32_t i = 0; i < n_seq; i++) {
                const uint64_t L = h_offsets[i + 1] - h_offsets[i];
                if (L >= (uint64_t) p->kmer_size && L - p->kmer_size + 1 > LONG_SEQ_KMERS) long_seqs.push_back(i);
            }
            skip_longer = LONG_SEQ_KMERS;
        }
    }
    KMU_TRY(launch_pmh3a(ctx, p, ds, d_block_rows, d_sig, nullptr, d_err, nullptr, 0, nullptr, nullptr, skip_longer, len_stats));
    for (uint32_t i : long_seqs) {
        DevSeqs one = ds;
        one.offsets = ds.offsets + i;
        one.packed_offsets = ds.packed_offsets ? ds.packed_offsets + i : nullptr;
        one.n_seq = 1;
        void *koff, *hk;
        KMU_TRY(dev_buf(ctx, "all.koff", 2 * 8 + 64, &koff));
        hipLaunchKernelGGL(k_nk_scan, dim3(1), dim3(1024), 0, ctx->stream, one.offsets, 1u, p->kmer_size, (uint64_t *) koff, d_err);
        uint64_t n_items = 0;
        KMU_HIP(ctx, hipMemcpyAsync(&n_items, (uint64_t *) koff + 1, 8, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        KMU_TRY(dev_buf(ctx, "all.hashes", n_items * 8 + 64, &hk));
        KmerCfg cfg{p->kmer_type, p->kmer_size, p->fhash};
        {
            KernelTimer t(ctx, "k_seq_hashes_compact");
            hipLaunchKernelGGL(k_seq_hashes_compact, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, one.bases, one.offsets,
                               one.packed_offsets, 1u, one.packed, one.total_bytes, cfg, (const uint64_t *) koff,
                               (uint64_t *) hk, d_err, 1);
        }
        KMU_HIP(ctx, hipGetLastError());
        KMU_TRY(sketch_all_hashed(ctx, p, (const uint64_t *) hk, n_items,
                                  reinterpret_cast<uint8_t *>(d_sig) + (size_t) i * p->sketch_size * sigb, d_err));
    }
    return KMU_OK;
}

// ProbMinHash3 (sketch_probminhash3, seqsketchjaccard.rs:272-319) generates the same points per key as ProbMinHash3a
// and keeps the same per-slot minimum: it runs on the ProbMinHash3a kernel (whole sequences only, like upstream).
static int resolve_algo(kmu_ctx *ctx, const kmu_sketch_params *p_in, kmu_sketch_params *p) {
    *p = *p_in;
    if (p->algo == KMU_ALGO_PROB3) {
        if (p->block_size > 0) return fail(ctx, KMU_E_UNSUPPORTED, "block sketching is ProbMinHash3a per sequence (seqblocksketch.rs:97)");
        p->algo = KMU_ALGO_PROB3A;
    }
    return KMU_OK;
}

extern "C" int kmu_sketch(kmu_ctx *ctx, const kmu_sketch_params *p_in, const uint8_t *bases, const uint64_t *offsets,
                          const uint64_t *packed_offsets, uint32_t n_seq, const uint64_t *block_row_offsets,
                          void *sig_out, uint32_t *counts_out) {
    if (!ctx || !p_in || !sig_out) return KMU_E_BAD_ARG;
    kmu_sketch_params p_res;
    KMU_TRY(resolve_algo(ctx, p_in, &p_res));
    const kmu_sketch_params *p = &p_res;
    KMU_TRY(check_kmer(ctx, p->kmer_type, p->kmer_size));
    KMU_TRY(sketch_params_check(ctx, p));
    if (!fhash_valid(p->fhash, p->kmer_type)) return fail(ctx, KMU_E_BAD_ARG, "fhash %d not valid for kmer_type %d", p->fhash, p->kmer_type);
    if (p->input_kind == KMU_INPUT_PACKED2 && (kmer_is_aa(p->kmer_type) || p->fhash == KMU_FHASH_CANON_NTHASH_8B))
        return fail(ctx, KMU_E_BAD_ARG, "packed input not valid for this kmer_type / fhash");
    if (p->mode == KMU_MODE_ALL_SEQS && p->algo == KMU_ALGO_BOTTOMK)
        return fail(ctx, KMU_E_UNSUPPORTED, "the reference has no bottom-k sketch over a list of sequences");
    if (p->block_size > 0 && !block_row_offsets) return fail(ctx, KMU_E_BAD_ARG, "block mode needs block_row_offsets");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, packed_offsets, n_seq, p->input_kind, p->mem, &ds));
    const size_t sigb = p->sig_type == KMU_SIG_U16 ? 2 : (p->sig_type == KMU_SIG_U32 || p->sig_type == KMU_SIG_F32) ? 4 : 8;
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
    if (p->mode == KMU_MODE_ALL_SEQS) {
        rows = 1;
        if (p->mem == KMU_MEM_HOST) {
            void *q;
            KMU_TRY(dev_buf(ctx, "out.sig", (size_t) p->sketch_size * sigb + 64, &q));
            d_sig = q;
        }
        if (algo_is_dens(p->algo)) { // one set of bins for every sequence, filled in place (setsketchert.rs:429-463)
            KMU_TRY(launch_dens(ctx, p, ds, d_sig, d_err, nullptr, 0));
            if (p->mem == KMU_MEM_HOST)
                KMU_HIP(ctx, hipMemcpyAsync(sig_out, d_sig, (size_t) p->sketch_size * sigb, hipMemcpyDeviceToHost, ctx->stream));
            if (!(p->mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
            return finish_call(ctx, p->mem);
        }
        // sketch_compressedkmer_seqs (setsketchert.rs:160-202, :299-335): one multiset / one stream over every
        // sequence.  fhash values of all k-mers, compact; then the pre-hashed all-sequences path.
        void *koff, *hk;
        KMU_TRY(dev_buf(ctx, "all.koff", ((size_t) n_seq + 1) * 8, &koff));
        hipLaunchKernelGGL(k_nk_scan, dim3(1), dim3(1024), 0, ctx->stream, ds.offsets, n_seq, p->kmer_size, (uint64_t *) koff, d_err);
        uint64_t n_items = 0;
        KMU_HIP(ctx, hipMemcpyAsync(&n_items, (uint64_t *) koff + n_seq, 8, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        KMU_TRY(dev_buf(ctx, "all.hashes", n_items * 8 + 64, &hk));
        if (n_seq) {
            KmerCfg cfg{p->kmer_type, p->kmer_size, p->fhash};
            const int spread = n_seq < (uint32_t) ctx->num_cus * 4 ? 1 : 0;
            int grid = spread ? ctx->num_cus * 8 : (int) std::min<uint32_t>(n_seq, (uint32_t) ctx->num_cus * 8);
            KernelTimer t(ctx, "k_seq_hashes_compact");
            hipLaunchKernelGGL(k_seq_hashes_compact, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets,
                               ds.packed_offsets, n_seq, ds.packed, ds.total_bytes, cfg, (const uint64_t *) koff,
                               (uint64_t *) hk, d_err, spread);
        }
        KMU_HIP(ctx, hipGetLastError());
        KMU_TRY(sketch_all_hashed(ctx, p, (const uint64_t *) hk, n_items, d_sig, d_err));
    } else if (n_seq) {
        switch (p->algo) {
        case KMU_ALGO_PROB3A: KMU_TRY(sketch_pmh_per_seq(ctx, p, ds, d_block_rows, d_sig, d_err, nullptr)); break;
        case KMU_ALGO_SUPER:
        case KMU_ALGO_SUPER2: KMU_TRY(launch_super(ctx, p, ds, d_sig, d_err, nullptr, 0, nullptr)); break;
        case KMU_ALGO_OPTDENS:
        case KMU_ALGO_REVOPTDENS:
        case KMU_ALGO_HLL: KMU_TRY(launch_dens(ctx, p, ds, d_sig, d_err, nullptr, 0)); break;
        case KMU_ALGO_BOTTOMK: KMU_TRY(launch_pmh3a(ctx, p, ds, nullptr, d_sig, d_counts, d_err)); break;
        default: return fail(ctx, KMU_E_UNSUPPORTED, "no per-sequence kernel for algo %d", p->algo);
        }
    }
    if (p->mem == KMU_MEM_HOST) {
        KMU_HIP(ctx, hipMemcpyAsync(sig_out, d_sig, rows * p->sketch_size * sigb, hipMemcpyDeviceToHost, ctx->stream));
        if (counts_out && p->mode != KMU_MODE_ALL_SEQS)
            KMU_HIP(ctx, hipMemcpyAsync(counts_out, d_counts, rows * p->sketch_size * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!(p->mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
    return finish_call(ctx, p->mem);
}

extern "C" int kmu_sketch_hashed(kmu_ctx *ctx, const kmu_sketch_params *p_in, const void *hashed, const uint64_t *offsets,
                                 uint32_t n_seq, void *sig_out, uint32_t *counts_out) {
    if (!ctx || !p_in || !sig_out || !offsets || (!hashed && n_seq)) return KMU_E_BAD_ARG;
    kmu_sketch_params p_res;
    KMU_TRY(resolve_algo(ctx, p_in, &p_res));
    const kmu_sketch_params *p = &p_res;
    KMU_TRY(sketch_params_check(ctx, p));
    if (p->block_size > 0) return fail(ctx, KMU_E_UNSUPPORTED, "block sketching needs the sequences (k-mer positions)");
    if (p->mode == KMU_MODE_ALL_SEQS && p->algo == KMU_ALGO_BOTTOMK)
        return fail(ctx, KMU_E_UNSUPPORTED, "the reference has no bottom-k sketch over a list of sequences");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    const int w = kmer_val_bytes(p->kmer_type);
    const size_t sigb = p->sig_type == KMU_SIG_U16 ? 2 : (p->sig_type == KMU_SIG_U32 || p->sig_type == KMU_SIG_F32) ? 4 : 8;
    const int m = p->sketch_size;
    const uint64_t rows = p->mode == KMU_MODE_ALL_SEQS ? 1 : n_seq;
    const void *d_vals = hashed;
    const uint64_t *d_off = offsets;
    void *d_sig = sig_out;
    uint32_t *d_counts = counts_out;
    uint64_t n_items = 0, first_item = 0; // the values of the sequences are hashed[offsets[0] .. offsets[n_seq])
    if (p->mem == KMU_MEM_HOST) {
        n_items = offsets[n_seq];
        first_item = n_seq ? offsets[0] : 0;
        void *q;
        KMU_TRY(dev_buf(ctx, "in.bases", n_items * w + 64, &q));
        KMU_HIP(ctx, hipMemcpyAsync(q, hashed, n_items * w, hipMemcpyHostToDevice, ctx->stream));
        d_vals = q;
        KMU_TRY(dev_buf(ctx, "in.offsets", ((size_t) n_seq + 1) * 8, &q));
        KMU_HIP(ctx, hipMemcpyAsync(q, offsets, ((size_t) n_seq + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        d_off = (const uint64_t *) q;
        KMU_TRY(dev_buf(ctx, "out.sig", rows * m * sigb + 64, &q));
        d_sig = q;
        if (counts_out) {
            KMU_TRY(dev_buf(ctx, "out.counts", rows * m * 4 + 64, &q));
            d_counts = (uint32_t *) q;
        }
    } else {
        KMU_HIP(ctx, hipMemcpyAsync(&n_items, offsets + n_seq, 8, hipMemcpyDeviceToHost, ctx->stream));
        if (n_seq) KMU_HIP(ctx, hipMemcpyAsync(&first_item, offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    if (algo_is_dens(p->algo)) {
        DevSeqs ds;
        ds.bases = reinterpret_cast<const uint8_t *>(d_vals);
        ds.offsets = d_off;
        ds.n_seq = n_seq;
        ds.total_bytes = 1;
        KMU_TRY(launch_dens(ctx, p, ds, d_sig, d_err, d_vals, w));
    } else if (p->mode == KMU_MODE_ALL_SEQS) {
        n_items -= first_item;
        const uint64_t *v64 = (const uint64_t *) d_vals + first_item;
        if (w == 4) { // widen to u64 once (the all-sequences path partitions u64 keys)
            void *q;
            KMU_TRY(dev_buf(ctx, "all.hashes", n_items * 8 + 64, &q));
            hipLaunchKernelGGL(k_widen_u32, dim3((unsigned) std::min<uint64_t>((n_items + 255) / 256 + 1, 65535)), dim3(256), 0,
                               ctx->stream, (const uint32_t *) d_vals + first_item, n_items, (uint64_t *) q);
            v64 = (const uint64_t *) q;
        }
        KMU_TRY(sketch_all_hashed(ctx, p, v64, n_items, d_sig, d_err));
    } else if (n_seq) {
        DevSeqs ds;
        ds.bases = reinterpret_cast<const uint8_t *>(d_vals);
        ds.offsets = d_off;
        ds.n_seq = n_seq;
        ds.total_bytes = 1;
        if (p->algo == KMU_ALGO_SUPER || p->algo == KMU_ALGO_SUPER2) KMU_TRY(launch_super(ctx, p, ds, d_sig, d_err, d_vals, w, nullptr));
        else KMU_TRY(launch_pmh3a(ctx, p, ds, nullptr, d_sig, d_counts, d_err, d_vals, w));
    }
    if (p->mem == KMU_MEM_HOST) {
        KMU_HIP(ctx, hipMemcpyAsync(sig_out, d_sig, rows * m * sigb, hipMemcpyDeviceToHost, ctx->stream));
        if (counts_out && p->mode != KMU_MODE_ALL_SEQS)
            KMU_HIP(ctx, hipMemcpyAsync(counts_out, d_counts, rows * m * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!(p->mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
    return finish_call(ctx, p->mem);
}

// per-sequence signatures of device-resident sequences for any algorithm (the switch of kmu_sketch)
static int sketch_per_seq_device(kmu_ctx *ctx, const kmu_sketch_params *p, const DevSeqs &ds, void *d_sig, uint32_t *d_err,
                                 const uint64_t *h_offsets) {
    switch (p->algo) {
    case KMU_ALGO_PROB3A: return sketch_pmh_per_seq(ctx, p, ds, nullptr, d_sig, d_err, h_offsets);
    case KMU_ALGO_SUPER:
    case KMU_ALGO_SUPER2: return launch_super(ctx, p, ds, d_sig, d_err, nullptr, 0, nullptr);
    case KMU_ALGO_OPTDENS:
    case KMU_ALGO_REVOPTDENS:
    case KMU_ALGO_HLL: return launch_dens(ctx, p, ds, d_sig, d_err, nullptr, 0);
    default: return fail(ctx, KMU_E_UNSUPPORTED, "kmu_sketch_count: no per-sequence kernel for algo %d", p->algo);
    }
}

extern "C" int kmu_sketch_count(kmu_ctx *ctx, const kmu_sketch_params *p_in, kmu_counter *counter, const uint8_t *bases,
                                const uint64_t *offsets, uint32_t n_seq, void *sig_out) {
    if (!ctx || !p_in || !sig_out || !offsets) return KMU_E_BAD_ARG;
    kmu_sketch_params p_res;
    KMU_TRY(resolve_algo(ctx, p_in, &p_res));
    const kmu_sketch_params *p = &p_res;
    KMU_TRY(check_kmer(ctx, p->kmer_type, p->kmer_size));
    KMU_TRY(sketch_params_check(ctx, p));
    if (!fhash_valid(p->fhash, p->kmer_type)) return fail(ctx, KMU_E_BAD_ARG, "fhash %d not valid for kmer_type %d", p->fhash, p->kmer_type);
    if (p->mode != KMU_MODE_PER_SEQ || p->block_size != 0 || p->input_kind != KMU_INPUT_ASCII || p->algo == KMU_ALGO_BOTTOMK)
        return fail(ctx, KMU_E_UNSUPPORTED, "kmu_sketch_count: whole unpacked sequences, one signature each");
    if (counter && counter_ctx(counter) != ctx) return fail(ctx, KMU_E_BAD_ARG, "the counter belongs to another context");
    if (counter && kmer_is_aa(p->kmer_type)) return fail(ctx, KMU_E_BAD_ARG, "counting is defined on DNA k-mers");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    const size_t sigb = p->sig_type == KMU_SIG_U16 ? 2 : (p->sig_type == KMU_SIG_U32 || p->sig_type == KMU_SIG_F32) ? 4 : 8;
    const size_t rowb = (size_t) p->sketch_size * sigb;
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    if (p->mem == KMU_MEM_DEVICE) {
        DevSeqs ds;
        KMU_TRY(stage_sequences(ctx, bases, offsets, nullptr, n_seq, KMU_INPUT_ASCII, KMU_MEM_DEVICE, &ds));
        if (counter) { // (a distributed counter: census, route, scatter, and the all-to-all leaves on the exchange stream)
            DevSeqs dc = ds;
            KMU_TRY(count_add_device_begin(counter, dc, nullptr, KMU_MEM_DEVICE, d_err));
        }
        if (n_seq) KMU_TRY(sketch_per_seq_device(ctx, p, ds, sig_out, d_err, nullptr));
        if (counter) KMU_TRY(count_add_device_end(counter));
        if (!ctx->async_device) KMU_TRY(check_err_word(ctx, d_err));
        return finish_call(ctx, KMU_MEM_DEVICE);
    }
    if (p->mem != KMU_MEM_HOST) return fail(ctx, KMU_E_BAD_ARG, "bad mem %d", p->mem);
    if (!bases && n_seq) return fail(ctx, KMU_E_BAD_ARG, "null sequence buffers");
    // ---- host buffers: upload | sketch | download | count as a pipeline over chunks of whole reads ----
    const uint64_t off0 = n_seq ? offsets[0] : 0, total = n_seq ? offsets[n_seq] - off0 : 0;
    std::vector<uint64_t> h_off((size_t) n_seq + 1);
    for (uint32_t i = 0; i <= n_seq; i++) h_off[i] = offsets[i] - off0;
    void *d_b, *d_o, *d_sig;
    KMU_TRY(dev_buf(ctx, "in.bases", total + 64, &d_b));
    KMU_TRY(dev_buf(ctx, "in.offsets", ((size_t) n_seq + 1) * 8, &d_o));
    KMU_TRY(dev_buf(ctx, "out.sig", (size_t) n_seq * rowb + 64, &d_sig));
    if (!ctx->pipe_h2d) KMU_HIP(ctx, hipStreamCreateWithFlags(&ctx->pipe_h2d, hipStreamNonBlocking));
    if (!ctx->pipe_d2h) KMU_HIP(ctx, hipStreamCreateWithFlags(&ctx->pipe_d2h, hipStreamNonBlocking));
    KMU_HIP(ctx, hipMemcpyAsync(d_o, h_off.data(), ((size_t) n_seq + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    uint64_t chunk_bytes = 512ull << 20;
    if (const char *e = getenv("KMU_PIPE_CHUNK_MB")) chunk_bytes = (uint64_t) std::max(1, atoi(e)) << 20;
    std::vector<uint32_t> cut(1, 0u); // chunk c = reads [cut[c], cut[c + 1])
    // Chunk sizes: the first one is an eighth of the others (the kernels start after 1 ms of upload instead of 9).  The upload
    // (4.38 GB at ~55 GB/s = 80 ms) is what the first phase is bound by -- a chunk's sketch + level 1 take 8 ms, its upload 9.3 --
    // and what is left when the last byte has arrived is the last chunk's sketch + level 1, then level 2 and the region build,
    // which need all of level 1.  (Tapering the last chunks shortens that tail by a chunk's sketch but pays for it in small
    // launches: 149.1 / 149.2 ms against 146.2 / 144.0 without, same box, r03: not kept.)
    // The bases cross PCIe packed (kmu_hostpack.hip): the host's cores pack chunk after chunk ahead of the upload, a quarter of the
    // bytes travels, a kernel on the upload stream restores the ASCII stream.  KMU_PIPE_PACK=0: the plain upload; small calls keep
    // it too.  Packed data arrive ~4x as fast as the kernels consume them, so the chunks may GROW: each three times its predecessor
    // (64 MB, 192 MB, 576 MB, 1.7 GB, the rest: every one is there before the kernels of the one before are through) -- five
    // launches of the sketch kernels instead of nine, each closer to the batched run's efficiency.  Headline workload, same box
    // (scripts/r04_hostleg.sh): 114.0 ms with growth 3, 115.3 / 116.0 with 2 / 4, 118.3 with equal chunks (KMU_PIPE_GROWTH=1),
    // 127.7 with the plain upload (KMU_PIPE_PACK=0); the device-resident step takes 104.5.
    bool packed_up = n_seq > 0 && total >= (32ull << 20);
    if (const char *e = getenv("KMU_PIPE_PACK")) packed_up = n_seq > 0 && atoi(e) != 0 && total >= 16;
    if (kmer_is_aa(p->kmer_type)) packed_up = false; // (residues are no bases: the 2-bit packer would reject every byte outside ACGT)
    uint64_t growth = packed_up ? 3 : 1;
    if (const char *e = getenv("KMU_PIPE_GROWTH")) growth = (uint64_t) std::max(1, atoi(e));
    std::vector<uint64_t> plan; // chunk sizes, in order (a chunk ends at the first read boundary at or behind its target)
    if (growth > 1) {
        uint64_t sz = std::min<uint64_t>(std::max<uint64_t>(chunk_bytes / 8, 1), total), left = total;
        while (left) {
            const uint64_t take = left <= sz + sz / 2 ? left : sz; // (a remainder of up to half a chunk more rides with the last one)
            plan.push_back(take);
            left -= take;
            sz *= growth;
        }
    } else {
        const uint64_t first = std::min<uint64_t>(std::max<uint64_t>(chunk_bytes / 8, 1), total);
        plan.push_back(first);
        const uint64_t body = total - first;
        const uint64_t n_body = std::max<uint64_t>(1, (body + chunk_bytes / 2) / chunk_bytes);
        for (uint64_t i = 0; i < n_body && body; i++) plan.push_back(body / n_body + 1);
    }
    {
        uint64_t target = 0;
        uint32_t r = 0;
        for (size_t i = 0; i < plan.size() && r < n_seq; i++) {
            target += plan[i];
            uint32_t e = i + 1 == plan.size() ? n_seq : (uint32_t) (std::lower_bound(h_off.begin() + r + 1, h_off.end(), target) - h_off.begin());
            if (e > n_seq) e = n_seq;
            if (e <= r) continue; // (a read that spans several targets: one chunk)
            cut.push_back(e);
            r = e;
        }
        if (r < n_seq) cut.push_back(n_seq);
    }
    const size_t n_chunks = cut.size() - 1;
    // The packed form of chunk c is the stream from the end of chunk c - 1 rounded up to 16 bases to its own end rounded up
    // likewise (the few bases of its first read before that came with the chunk before).
    packed_up = packed_up && n_chunks > 0;
    std::vector<uint64_t> pk_bounds;
    void *h_packed = nullptr, *d_packed = nullptr;
    std::unique_ptr<PackPipe> packer;
    if (packed_up) {
        pk_bounds.push_back(0);
        for (size_t c = 0; c < n_chunks; c++) {
            const uint64_t e = c + 1 == n_chunks ? total : std::min<uint64_t>(total, (h_off[cut[c + 1]] + 15) & ~15ull);
            pk_bounds.push_back(std::max(e, pk_bounds.back()));
        }
        KMU_TRY(host_buf(ctx, "pipe.packed", (size_t) (total / 4 + 64), &h_packed));
        KMU_TRY(dev_buf(ctx, "pipe.packed_d", (size_t) (total / 4 + 64), &d_packed));
        int threads = 16;
        threads = std::min<int>(threads, std::max(1u, std::thread::hardware_concurrency()));
        try {
            packer.reset(new PackPipe(bases + off0, (uint8_t *) h_packed, total, threads));
        } catch (const std::exception &) { // (no threads to be had: the plain upload, same chunks)
            packer.reset();
            packed_up = false;
        }
    }
    std::vector<hipEvent_t> ev_up(n_chunks), ev_sk(n_chunks);
    for (size_t c = 0; c < n_chunks; c++) {
        KMU_HIP(ctx, hipEventCreateWithFlags(&ev_up[c], hipEventDisableTiming));
        KMU_HIP(ctx, hipEventCreateWithFlags(&ev_sk[c], hipEventDisableTiming));
    }
    int rc = KMU_OK;
    // (the uploads of the packed form are enqueued by a thread of their own: nothing in here touches the context's error state --
    //  a failure comes back as a code and a text, and the calling thread reports it)
    auto upload = [&](size_t c, std::string *msg) -> int {
        auto hip_ok = [&](hipError_t e, const char *what) {
            if (e == hipSuccess) return true;
            *msg = std::string(what) + ": " + hipGetErrorString(e);
            return false;
        };
        if (packed_up) {
            // in pieces of 64 M bases, each as soon as it is packed: the chunk's packing runs under its own upload.  Only copies go
            // on the upload stream: the kernel that restores the ASCII stream runs on the compute stream in front of the chunk's
            // kernels (on the upload stream it would wait for a CU that the persistent sketch kernels do not release, and every
            // copy behind it with it)
            const uint64_t p0 = pk_bounds[c], p1 = pk_bounds[c + 1], piece = 64ull << 20;
            for (uint64_t q0 = p0; q0 < p1; q0 += piece) {
                const uint64_t q1 = std::min(p1, q0 + piece);
                if (!packer->wait_prefix(q1)) {
                    *msg = "pattern not a code in alphabet_2b (non-ACGT byte in a sequence)";
                    return KMU_E_NON_ACGT;
                }
                if (!hip_ok(hipMemcpyAsync((uint8_t *) d_packed + q0 / 4, (const uint8_t *) h_packed + q0 / 4, (size_t) ((q1 - q0 + 3) / 4), hipMemcpyHostToDevice,
                                           ctx->pipe_h2d), "upload of a packed chunk")) return KMU_E_HIP;
            }
            return hip_ok(hipEventRecord(ev_up[c], ctx->pipe_h2d), "hipEventRecord") ? KMU_OK : KMU_E_HIP;
        }
        const uint64_t b0 = h_off[cut[c]], b1 = h_off[cut[c + 1]];
        if (!hip_ok(hipMemcpyAsync((uint8_t *) d_b + b0, bases + off0 + b0, b1 - b0, hipMemcpyHostToDevice, ctx->pipe_h2d), "upload of a chunk")) return KMU_E_HIP;
        return hip_ok(hipEventRecord(ev_up[c], ctx->pipe_h2d), "hipEventRecord") ? KMU_OK : KMU_E_HIP;
    };
    DevSeqs all;
    all.bases = (const uint8_t *) d_b;
    all.offsets = (const uint64_t *) d_o;
    all.n_seq = n_seq;
    all.total_bytes = total;
    // the count's level-1 partition runs under the upload too, for the part of the stream that has arrived
    void *cc = nullptr;
    int cc_on = 0;
    if (counter && n_seq) {
        DevSeqs dc = all;
        rc = count_chunked_begin(counter, dc, h_off.data(), d_err, &cc, &cc_on);
    }
    // Packed uploads are enqueued by a thread of their own: an upload waits for the packer, and the thread that launches the
    // kernels of chunk c must not stand behind the packing of chunk c + 1.  It waits (on the host) only until the upload of ITS
    // chunk has been enqueued -- an event that has not been recorded yet cannot be waited for on a stream.
    std::mutex up_mu;
    std::condition_variable up_cv;
    size_t up_done = 0; // uploads of chunks [0, up_done) are enqueued
    int up_rc = KMU_OK;
    std::string up_msg, my_msg;
    std::thread uploader;
    if (packed_up && rc == KMU_OK) {
        try {
            uploader = std::thread([&] {
                (void) hipSetDevice(ctx->device);
                for (size_t c = 0; c < n_chunks; c++) {
                    std::string m;
                    const int r = upload(c, &m);
                    std::lock_guard<std::mutex> g(up_mu);
                    if (r != KMU_OK) { up_rc = r; up_msg = m; }
                    up_done = r == KMU_OK ? c + 1 : n_chunks; // (a failure releases every waiter)
                    up_cv.notify_all();
                    if (r != KMU_OK) return;
                }
            });
        } catch (const std::exception &) { rc = fail(ctx, KMU_E_HIP, "kmu_sketch_count: cannot start the upload thread"); }
    } else if (n_chunks && rc == KMU_OK) {
        rc = upload(0, &my_msg);
        if (rc != KMU_OK) (void) fail(ctx, rc, "%s", my_msg.c_str());
    }
    for (size_t c = 0; c < n_chunks && rc == KMU_OK; c++) {
        if (packed_up) {
            std::unique_lock<std::mutex> g(up_mu);
            up_cv.wait(g, [&] { return up_done > c; });
            rc = up_rc;
            if (rc != KMU_OK) (void) fail(ctx, rc, "%s", up_msg.c_str());
        } else if (c + 1 < n_chunks) {
            rc = upload(c + 1, &my_msg);
            if (rc != KMU_OK) (void) fail(ctx, rc, "%s", my_msg.c_str());
        }
        if (rc != KMU_OK) break;
        DevSeqs ds = all;
        ds.offsets = all.offsets + cut[c];
        ds.n_seq = cut[c + 1] - cut[c];
        uint8_t *d_rows = (uint8_t *) d_sig + (size_t) cut[c] * rowb;
        if (hipStreamWaitEvent(ctx->stream, ev_up[c], 0) != hipSuccess) { rc = fail(ctx, KMU_E_HIP, "hipStreamWaitEvent failed"); break; }
        if (packed_up && pk_bounds[c + 1] > pk_bounds[c]) {
            KernelTimer tm(ctx, "k_unpack2b");
            rc = launch_unpack2b(ctx, (const uint8_t *) d_packed + pk_bounds[c] / 4, pk_bounds[c + 1] - pk_bounds[c], (uint8_t *) d_b + pk_bounds[c], ctx->stream);
            if (rc != KMU_OK) break;
        }
        rc = sketch_per_seq_device(ctx, p, ds, d_rows, d_err, h_off.data() + cut[c]);
        if (rc != KMU_OK) break;
        if (hipEventRecord(ev_sk[c], ctx->stream) != hipSuccess || hipStreamWaitEvent(ctx->pipe_d2h, ev_sk[c], 0) != hipSuccess ||
            hipMemcpyAsync((uint8_t *) sig_out + (size_t) cut[c] * rowb, d_rows, (size_t) ds.n_seq * rowb, hipMemcpyDeviceToHost,
                           ctx->pipe_d2h) != hipSuccess)
            rc = fail(ctx, KMU_E_HIP, "signature download failed: %s", hipGetErrorString(hipGetLastError()));
        if (rc == KMU_OK && cc_on) rc = count_chunked_level1(counter, cc, h_off[cut[c + 1]]);
    }
    if (rc == KMU_OK && counter) { // the whole read set is resident by now: ev_up of the last chunk has been waited for
        if (cc_on) {
            rc = count_chunked_finish(counter, cc);
            cc = nullptr;
        } else {
            DevSeqs dc = all;
            rc = count_add_device_begin(counter, dc, h_off.data(), KMU_MEM_HOST, d_err);
            if (rc == KMU_OK) rc = count_add_device_end(counter);
        }
    }
    if (cc) count_chunked_abort(cc);
    if (uploader.joinable()) uploader.join(); // (a call that failed elsewhere: the uploads still queued are harmless, the buffers stay)
    packer.reset(); // (joins the workers: a call that ends early stops them at their next slab)
    (void) hipStreamSynchronize(ctx->pipe_h2d);
    (void) hipStreamSynchronize(ctx->pipe_d2h);
    for (size_t c = 0; c < n_chunks; c++) {
        (void) hipEventDestroy(ev_up[c]);
        (void) hipEventDestroy(ev_sk[c]);
    }
    if (rc != KMU_OK) {
        (void) hipStreamSynchronize(ctx->stream);
        return rc;
    }
    KMU_TRY(check_err_word(ctx, d_err));
    return finish_call(ctx, KMU_MEM_HOST);
}

// ---- one signature for sequences held by several GPUs ------------------------------------------------------------------
// Every rank turns ITS share into per-slot minima (a "partial"), the ranks exchange these small arrays (all-gather) and each
// merges them.  SuperMinHash / SuperMinHash2 / OptDens / RevOptDens sketch k-mer occurrences independently, so a rank's share
// is simply its sequences.  ProbMinHash weighs a key by its multiplicity over ALL sequences: the shares must hold disjoint
// key sets (exchange the hashed k-mers by owner first, as counting does), then per-slot (h, key) minima merge exactly.
extern "C" uint32_t kmu_sketch_partial_words(const kmu_sketch_params *p) {
    if (!p || p->sketch_size < 1) return 0;
    const bool prob = p->algo == KMU_ALGO_PROB3A || p->algo == KMU_ALGO_PROB3;
    return (uint32_t) p->sketch_size * (prob ? 2u : 1u);
}

static int partial_begin(kmu_ctx *ctx, const kmu_sketch_params *p_in, kmu_sketch_params *p, uint64_t *partial_out, uint64_t **d_part) {
    if (!ctx || !p_in || !partial_out) return KMU_E_BAD_ARG;
    *p = *p_in;
    p->mode = KMU_MODE_ALL_SEQS;
    p->block_size = 0;
    if (p->algo == KMU_ALGO_BOTTOMK) return fail(ctx, KMU_E_UNSUPPORTED, "the reference has no bottom-k sketch over a list of sequences");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    *d_part = partial_out;
    if (p->mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "out.partial", (size_t) kmu_sketch_partial_words(p) * 8 + 64, &q));
        *d_part = (uint64_t *) q;
    }
    return KMU_OK;
}

static int partial_end(kmu_ctx *ctx, const kmu_sketch_params *p, int rc, uint64_t *partial_out, const uint64_t *d_part) {
    ctx->partial_out = nullptr;
    if (rc != KMU_OK) return rc;
    if (p->mem == KMU_MEM_HOST) {
        KMU_HIP(ctx, hipMemcpyAsync(partial_out, d_part, (size_t) kmu_sketch_partial_words(p) * 8, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return KMU_OK;
}

extern "C" int kmu_sketch_partial(kmu_ctx *ctx, const kmu_sketch_params *p_in, const uint8_t *bases, const uint64_t *offsets,
                                  const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *partial_out) {
    kmu_sketch_params p;
    uint64_t *d_part = nullptr;
    KMU_TRY(partial_begin(ctx, p_in, &p, partial_out, &d_part));
    void *scratch_sig; // the signature slot of the inner call is not used
    KMU_TRY(dev_buf(ctx, "out.sig", (size_t) p.sketch_size * 8 + 64, &scratch_sig));
    std::vector<uint64_t> host_sig(p.mem == KMU_MEM_HOST ? (size_t) p.sketch_size : 0);
    ctx->partial_out = d_part;
    const int rc = kmu_sketch(ctx, &p, bases, offsets, packed_offsets, n_seq, nullptr,
                              p.mem == KMU_MEM_HOST ? (void *) host_sig.data() : scratch_sig, nullptr);
    return partial_end(ctx, &p, rc, partial_out, d_part);
}

extern "C" int kmu_sketch_hashed_partial(kmu_ctx *ctx, const kmu_sketch_params *p_in, const void *hashed, const uint64_t *offsets,
                                         uint32_t n_seq, uint64_t *partial_out) {
    kmu_sketch_params p;
    uint64_t *d_part = nullptr;
    KMU_TRY(partial_begin(ctx, p_in, &p, partial_out, &d_part));
    void *scratch_sig;
    KMU_TRY(dev_buf(ctx, "out.sig", (size_t) p.sketch_size * 8 + 64, &scratch_sig));
    std::vector<uint64_t> host_sig(p.mem == KMU_MEM_HOST ? (size_t) p.sketch_size : 0);
    ctx->partial_out = d_part;
    const int rc = kmu_sketch_hashed(ctx, &p, hashed, offsets, n_seq, p.mem == KMU_MEM_HOST ? (void *) host_sig.data() : scratch_sig,
                                     nullptr);
    return partial_end(ctx, &p, rc, partial_out, d_part);
}

extern "C" int kmu_sketch_merge_partials(kmu_ctx *ctx, const kmu_sketch_params *p_in, const uint64_t *partials, uint32_t n_parts,
                                         void *sig_out) {
    if (!ctx || !p_in || !partials || !sig_out || n_parts == 0) return KMU_E_BAD_ARG;
    kmu_sketch_params p_res;
    KMU_TRY(resolve_algo(ctx, p_in, &p_res));
    const kmu_sketch_params *p = &p_res;
    KMU_TRY(sketch_params_check(ctx, p));
    if (p->algo == KMU_ALGO_BOTTOMK) return fail(ctx, KMU_E_UNSUPPORTED, "no bottom-k sketch over a list of sequences");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    const int m = p->sketch_size;
    const size_t sigb = p->sig_type == KMU_SIG_U16 ? 2 : (p->sig_type == KMU_SIG_U32 || p->sig_type == KMU_SIG_F32) ? 4 : 8;
    const uint64_t words = kmu_sketch_partial_words(p);
    const uint64_t *d_parts = partials;
    void *d_sig = sig_out;
    if (p->mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "in.partials", (size_t) n_parts * words * 8 + 64, &q));
        KMU_HIP(ctx, hipMemcpyAsync(q, partials, (size_t) n_parts * words * 8, hipMemcpyHostToDevice, ctx->stream));
        d_parts = (const uint64_t *) q;
        KMU_TRY(dev_buf(ctx, "out.sig", (size_t) m * sigb + 64, &q));
        d_sig = q;
    }
    if (p->algo == KMU_ALGO_PROB3A) {
        hipLaunchKernelGGL(k_pmh_reduce, dim3(m), dim3(256), 0, ctx->stream, d_parts, d_parts + m, (uint64_t) n_parts, m, words,
                           kmer_val_bytes(p->kmer_type), d_sig, (uint64_t *) nullptr);
    } else if (algo_is_dens(p->algo)) {
        KMU_TRY(launch_dens_merge(ctx, p, d_parts, n_parts, d_sig));
    } else {
        KMU_TRY(launch_super_reduce(ctx, p, d_parts, n_parts, d_sig));
    }
    KMU_HIP(ctx, hipGetLastError());
    if (p->mem == KMU_MEM_HOST) KMU_HIP(ctx, hipMemcpyAsync(sig_out, d_sig, (size_t) m * sigb, hipMemcpyDeviceToHost, ctx->stream));
    return finish_call(ctx, p->mem);
}

// fhash of every k-mer of every sequence, one after the other (no gaps for the positions that start no k-mer): what a rank
// hands to the owner exchange of a distributed ProbMinHash sketch.  out == NULL: only the number of values in *n_out.
extern "C" int kmu_kmer_hashes_compact(kmu_ctx *ctx, const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                                       const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *out, uint64_t cap, uint64_t *n_out) {
    if (!ctx || !p || !n_out) return KMU_E_BAD_ARG;
    KMU_TRY(check_kmer(ctx, p->kmer_type, p->kmer_size));
    if (!fhash_valid(p->fhash, p->kmer_type)) return fail(ctx, KMU_E_BAD_ARG, "fhash %d not valid for kmer_type %d", p->fhash, p->kmer_type);
    if (p->input_kind == KMU_INPUT_PACKED2 && (kmer_is_aa(p->kmer_type) || p->fhash == KMU_FHASH_CANON_NTHASH_8B))
        return fail(ctx, KMU_E_BAD_ARG, "packed input not valid for this kmer_type / fhash");
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    *n_out = 0;
    if (n_seq == 0) return KMU_OK;
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, packed_offsets, n_seq, p->input_kind, p->mem, &ds));
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    void *koff;
    KMU_TRY(dev_buf(ctx, "all.koff", ((size_t) n_seq + 1) * 8, &koff));
    hipLaunchKernelGGL(k_nk_scan, dim3(1), dim3(1024), 0, ctx->stream, ds.offsets, n_seq, p->kmer_size, (uint64_t *) koff, d_err);
    uint64_t n_items = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&n_items, (uint64_t *) koff + n_seq, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n_items;
    if (!out) return KMU_OK;
    if (cap < n_items) return fail(ctx, KMU_E_BAD_ARG, "output too small: %llu values", (unsigned long long) n_items);
    uint64_t *d_out = out;
    if (p->mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "all.hashes", n_items * 8 + 64, &q));
        d_out = (uint64_t *) q;
    }
    {
        KmerCfg cfg{p->kmer_type, p->kmer_size, p->fhash};
        const int spread = n_seq < (uint32_t) ctx->num_cus * 4 ? 1 : 0;
        const int grid = spread ? ctx->num_cus * 8 : (int) std::min<uint32_t>(n_seq, (uint32_t) ctx->num_cus * 8);
        KernelTimer t(ctx, "k_seq_hashes_compact");
        hipLaunchKernelGGL(k_seq_hashes_compact, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets, ds.packed_offsets,
                           n_seq, ds.packed, ds.total_bytes, cfg, (const uint64_t *) koff, d_out, d_err, spread);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (p->mem == KMU_MEM_HOST) KMU_HIP(ctx, hipMemcpyAsync(out, d_out, n_items * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (!(p->mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
    return finish_call(ctx, p->mem);
}
