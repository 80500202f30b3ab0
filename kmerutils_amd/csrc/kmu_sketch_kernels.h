// kmu_sketch_kernels.h -- what the host side of the sketch path (kmu_sketch.hip: routes, launches, the C entry points) needs of
// the per-sequence sketch kernels (kmu_sketch_kernels.hip): their argument block, the constants their LDS budgets are made of,
// and their declarations (the template kernels are instantiated in kmu_sketch_kernels.hip for exactly the forms listed here).
#pragma once

#include "kmu_ctx.hpp"
#include "kmu_device.h"

namespace kmu {

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
    uint32_t cap;         // dense key / weight capacity of one pass
    uint32_t part_target; // k-mers aimed at per pass (cap minus a fluctuation margin)
    double inv_part_target;
    uint32_t tile_shift;  // log2 of the k-mer positions covered by one staged tile
    uint64_t *scr_keys;   // per-workgroup overflow scratch (keys that found no register slot): [grid][cap]
    uint32_t *scr_info;   //   (bucket << 16) | rank-or-position
    uint32_t *scr_w;      //   weight carried in (bottom-k running list), 1 otherwise
    uint32_t skip_longer; // != 0: sequences with more k-mers than this are left to the global (partitioned) path
    uint64_t *def_keys;   // per-workgroup list of the keys that belong to later partition passes: [grid][DEF_CAP]
    // bottom-k (MinHashCount, src/sketching/minhash.rs:62-99)
    int bk_shift;         // bucket = (key >> bk_shift) & 0xFFF: the 12 most significant *used* bits of the hash
    uint32_t bk_mask;     // count wrap mask: 0xFFFF (u16 counts) or 0xFF (MinInvHashCountKmer)
    uint64_t *bk_keys;    // per-workgroup running list between partition passes: [grid][m]
    uint32_t *bk_cnt;
    uint32_t *counts_out; // may be null
    // pre-hashed input (kmu_sketch_hashed, and the leaves of a sketch over all sequences): the "sequence" is an array
    // of Kmer::Val values, offsets count values; runs on the AA instantiation (no code-word staging) with k = 1
    const void *hashed;
    int hashed_bytes;     // 0 = sequences of bases / residues, 4 / 8 = width of the pre-hashed values
    uint64_t *part_h;     // non-null: write the slot minima (h bits, arg-min key) of every "sequence" here instead
    uint64_t *part_k;     //   of a signature row: partial results of disjoint key sets, merged by k_pmh_reduce
    // two-kernel ProbMinHash3a (whole sequences): the multiset kernel leaves the distinct (key, weight) pairs of read r
    // in lst_keys / lst_w [offsets[r] .. offsets[r] + lst_n[r]); k_pmh_points turns them into the signature row
    uint64_t *lst_keys;
    uint32_t *lst_w;
    uint32_t *lst_n;
    // the first lst_nu[r] entries of read r's list have weight 1 and NO entry in lst_w (k_multiset_uq: the keys proven to occur
    // once, nine in ten of an ONT read, leave as 8 bytes instead of 12 and their weights are never read back); zeroed by the
    // host before the multiset kernels, only k_multiset_uq writes it
    uint32_t *lst_nu;
    uint32_t *queue2;     // read counter of k_pmh_points
    uint32_t *pts_long;   // k_pmh_points: reads a whole workgroup takes ([0] count, [2..] indices; k_pts_long_list), or null
    uint32_t pts_long_t;  // ... those with more list entries than this
    // The PLAIN instantiation leaves a sequence whose k-mers overflow a pass (repetitive reads: rounds with carry lists) to
    // the general one: it appends the sequence to redo_list (count in queue[56]); the second launch walks read_list.
    uint32_t *redo_list;
    const uint32_t *read_list; // non-null: queue entry q stands for sequence read_list[q]
    uint32_t n_queue;          // queue entries (n_seq, or the length of read_list)
    uint32_t tile_words;  // staged code words per tile (16 bases each)
    uint32_t idx_thresh;  // rand 0.9 Uniform<usize>(0, m): reject while lo < (2^32 - m) % m
    uint64_t idx_zone;    // rand 0.8 Uniform<usize>(0, m): accept while lo <= zone
    uint32_t ablate;      // diagnostics only (KMU_PMH_ABLATE): 1 skip pass B math, 2 skip table insert, 4 skip hashing
    Exp01 e01;
    void *sig_out;
    uint32_t *queue; // atomic read counter
    uint32_t *err;
};

// misc words in LDS
enum { M_READ = 0, M_NSCR = 1, M_NEXT = 2, M_DEF = 3, M_QMAX = 4 /* 4,5: u64 */, M_FLAGS = 6 /* 6,7 */, M_WORDS = 8 };
static constexpr uint32_t DEF_CAP = 1u << 20;       // keys of later partitions a workgroup can set aside per block,
static constexpr uint32_t DEF_PARTS = 32;           //   one sub-list of DEF_CAP / DEF_PARTS keys per later partition
static constexpr uint32_t DEF_SEG = DEF_CAP / DEF_PARTS;
static constexpr uint32_t LONG_SEQ_KMERS = 1u << 18; // longer sequences take the global partitioned route (kmu_sketch)
static constexpr int BUCKET_BITS = 12;               // counting-sort buckets
static constexpr uint32_t NBUCKETS = 1u << BUCKET_BITS;
static constexpr int QCHUNK = 4;                     // reads taken from the queue per atomic
static constexpr int KREG = 10;                      // keys a thread keeps in registers between the sort phases
// 1 / w.  Most weights are tiny: with a table of the exact quotients (filled once per workgroup with the same IEEE
// division) the 25-instruction f64 division only runs for the rare wave that holds a weight beyond the table.
static constexpr uint32_t WINV_LUT = 256;
// k_pmh_points, u64 words per wave beside the 2 m slot words: q_max (2), queue of 128 keys, weights (64), state words (256)
static constexpr size_t PTS_WAVE_WORDS = 2 + 128 + 64 + 256;
static constexpr int UQ_KREG = 20;
// The two shapes' bitmap sizes (log2 bits) and collected-key capacities.  Round 4: bitmaps twice as large (nearly every key of a
// collision group of an ONT read is a false positive of the bitmap: 9 % of the keys at 2^16 bits, 4.5 % at 2^17), collected-key
// arrays half as large to pay for them in LDS: the sketch unit 50.3 -> 49.6 ms, same rows.
static constexpr uint32_t UQ1_BM = 17, UQ1_COLL = 1024, UQ2_BM = 18, UQ2_COLL = 2048;
template <int UQ_THREADS, uint32_t UQ_BM_BITS, uint32_t UQ_COLL>
struct UqShape {
    static constexpr uint32_t KEYS = (uint32_t) UQ_THREADS * UQ_KREG;
    static constexpr uint32_t BM_WORDS = (1u << UQ_BM_BITS) / 32;
    static constexpr uint32_t BUCKETS = 2u * UQ_THREADS; // of the collision groups' counting sort: two per thread
    static constexpr uint32_t TILE = (KEYS + 32 + 15) / 16 + 3; // staged code words of a read
    static constexpr uint32_t RAW_WAVES = (TILE + 63) / 64; // landing area of the next read's 16-byte chunks: 1 KiB per wave instruction
    static constexpr size_t LDS = (size_t) BM_WORDS * 8 + (size_t) UQ_COLL * 20 + ((size_t) BUCKETS + 1 + TILE + UQ_THREADS / 64 + 8 + 8) * 4 + 64 +
                                  (size_t) RAW_WAVES * 1024 + 16;
};
static constexpr uint32_t SHORT_KEYS = 256, SHORT_SLOTS = 512, SHORT_WORDS = 24;
static constexpr size_t SHORT_WAVE_BYTES = (size_t) SHORT_SLOTS * 12 + SHORT_WORDS * 4;
static constexpr uint32_t SMALLK_WORDS = 32768;  // LDS words of the histogram (128 KiB)
static constexpr uint32_t SMALLK_TILE = 1024;    // staged code words per tile

// diagnostic builds (-DKMU_DIAG=1): bits of KMU_PMH_ABLATE leave phases of the kernels out / clock them
#define ABL(bits) (KMU_DIAG && (a.ablate & (bits)))

// ---- the kernels (definitions and comments: kmu_sketch_kernels.hip) ------------------------------------------------------------
template <bool AA, bool BOTTOMK, bool EMIT = false, bool PLAIN = false>
__global__ void k_sketch_pmh3a(SketchArgs a);
__global__ void k_pts_long_list(const uint32_t *lst_n, uint32_t n_seq, uint32_t thr, uint32_t *out);
template <bool SIG32>
__global__ void k_pmh_points(SketchArgs a);
template <int UQ_THREADS, uint32_t UQ_BM_BITS, uint32_t UQ_COLL, int MINW>
__global__ void k_multiset_uq(SketchArgs a);
__global__ void k_multiset_short(SketchArgs a);
template <bool SIG32>
__global__ void k_pmh_points_short(SketchArgs a);
template <bool EMIT>
__global__ void k_sketch_smallk(SketchArgs a);
__global__ void k_pmh_reduce(const uint64_t *part_h, const uint64_t *part_k, uint64_t n_parts, int m, uint64_t stride, int sig_bytes, void *sig_out,
                             uint64_t *part_out);
__global__ void k_max_len(const uint64_t *offsets, uint32_t n_seq, uint64_t *out);
__global__ void k_nk_scan(const uint64_t *offsets, uint32_t n_seq, int k, uint64_t *koff, uint32_t *err);
__global__ void k_seq_hashes_compact(const uint8_t *bases, const uint64_t *offsets, const uint64_t *packed_offsets, uint32_t n_seq, int packed,
                                     uint64_t total, KmerCfg cfg, const uint64_t *koff, uint64_t *out, uint32_t *err, int spread);
__global__ void k_widen_u32(const uint32_t *in, uint64_t n, uint64_t *out);

// the forms of the template kernels the host launches: instantiated in kmu_sketch_kernels.hip, declared for everybody else
#define KMU_SKETCH_KERNEL_FORMS(X)                                                                                                  \
    X(k_sketch_pmh3a<false, false>) X(k_sketch_pmh3a<false, false, true>) X(k_sketch_pmh3a<true, true>) X(k_sketch_pmh3a<true, false>) \
    X(k_sketch_pmh3a<false, true>) X(k_sketch_pmh3a<false, false, true, true>) X(k_sketch_pmh3a<false, false, false, true>)            \
    X(k_pmh_points<true>) X(k_pmh_points<false>) X(k_pmh_points_short<true>) X(k_pmh_points_short<false>)                             \
    X(k_multiset_uq<512, UQ1_BM, UQ1_COLL, 4>) X(k_multiset_uq<1024, UQ2_BM, UQ2_COLL, 4>) X(k_sketch_smallk<true>) X(k_sketch_smallk<false>)
#ifndef KMU_SKETCH_KERNELS_TU
#define KMU_X_EXTERN(...) extern template __global__ void __VA_ARGS__(SketchArgs);
KMU_SKETCH_KERNEL_FORMS(KMU_X_EXTERN)
#undef KMU_X_EXTERN
#endif

} // namespace kmu
