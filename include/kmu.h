/*
 * kmu.h -- C-ABI boundary of the MI355X (gfx950) k-mer generation + counting + sketching hot path.
 *
 * This is the drop-in boundary for the hot path of jean-pierreBoth/kmerutils (a Rust crate with no FFI of its
 * own): every entry point below replaces a *Rust* interface of the reference, cited as `file:line` relative
 * to the reference tree.  A Rust closure cannot cross an FFI, so the `fhash` closures the reference actually
 * uses are enumerated in `kmu_fhash`.  INTEGRATION.md shows the `extern "C"` block and the
 * `impl SeqSketcherT<Kmer> for ...` shim a maintainer would add on the Rust side.
 *
 * Conventions
 *  - plain C scalars / pointers / sizes only; no torch, no C++ types.
 *  - every call returns 0 (KMU_OK) or a negative kmu_status; kmu_last_error() gives the text.
 *  - the caller owns every input and output buffer.  `mem` says where they live:
 *      KMU_MEM_HOST   : host pointers; the library stages H2D / D2H itself (synchronous call).
 *      KMU_MEM_DEVICE : HIP device pointers (e.g. a torch tensor's data_ptr()); kernels are enqueued on the
 *                       context's stream and the call returns after the stream has been synchronised unless
 *                       the context was created with `async_device = 1`.  Such calls cannot report what their kernels
 *                       find (a non-ACGT byte, an empty sequence, a full table): the device error word is sticky, and
 *                       the bits of every call since the last synchronising call come back from the next one --
 *                       kmu_synchronize at the latest.
 *  - sequences are handed over as one concatenated byte array + (n_seq+1) uint64 offsets:
 *      KMU_INPUT_ASCII  : bases[offsets[i] .. offsets[i+1]) are the ASCII letters of sequence i
 *      KMU_INPUT_PACKED2: offsets are in *bases*; sequence i starts at byte `packed_byte_offsets[i]`
 *                         and is packed exactly as reference `Sequence::new(raw,2)`
 *                         (src/base/sequence.rs:48-73: 4 bases / byte, first base in bits 7..6).
 *    offsets[0] need not be 0: `offsets + first` with n_seq = last - first names the reads [first, last) of a larger
 *    array without copying anything (the base pointer stays the array's).
 *  - a kmu_ctx is used from one thread at a time; distinct contexts may run concurrently.
 */
#ifndef KMU_H
#define KMU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMU_VERSION_MAJOR 0
#define KMU_VERSION_MINOR 1

typedef struct kmu_ctx kmu_ctx;
typedef struct kmu_counter kmu_counter;

typedef enum kmu_status {
    KMU_OK = 0,
    KMU_E_BAD_K = -1,        /* reference: panic in KmerSeqIterator::new, src/base/kmergenerator.rs:48-53 */
    KMU_E_BAD_ALPHABET = -2,
    KMU_E_EMPTY_SEQ = -3,    /* reference: ilog2(0) panic, src/sketching/nbkmerguess.rs:8 */
    KMU_E_NON_ACGT = -4,     /* reference: Alphabet2b::encode panics, src/base/alphabet.rs:125 */
    KMU_E_OOM = -5,
    KMU_E_HIP = -6,
    KMU_E_BAD_ARG = -7,
    KMU_E_TABLE_FULL = -8,
    KMU_E_UNSUPPORTED = -9,
    KMU_E_NO_DEVICE = -10,
    KMU_E_RCCL = -11         /* RCCL (or the host's transport) reported an error; text in kmu_last_error */
} kmu_status;

typedef enum kmu_mem { KMU_MEM_HOST = 0, KMU_MEM_DEVICE = 1 } kmu_mem;
typedef enum kmu_input_kind { KMU_INPUT_ASCII = 0, KMU_INPUT_PACKED2 = 1 } kmu_input_kind;

/* k-mer value types of the reference (src/base/kmer32bit.rs:22, kmer16b32bit.rs:21, kmer64bit.rs:24,
 * src/aautils/kmeraa.rs:146,280).  The type fixes Kmer::Val (u32 / u64) and the bit layout of `.0`. */
typedef enum kmu_kmer_type {
    KMU_KMER32BIT = 0,    /* k <= 14, u32, k stored in bits 31..28            */
    KMU_KMER16B32BIT = 1, /* k == 16, u32                                     */
    KMU_KMER64BIT = 2,    /* k <= 31, (u64, u8); k = 32 is broken upstream    */
    KMU_KMERAA32BIT = 3,  /* amino acids, 5 bits, k <= 6, u32                 */
    KMU_KMERAA64BIT = 4   /* amino acids, 5 bits, k <= 12, u64                */
} kmu_kmer_type;

/* The closures `fhash: Fn(&Kmer) -> Kmer::Val` that reference callers really pass. */
typedef enum kmu_fhash {
    KMU_FHASH_IDENTITY_RAW = 0,   /* kmer.0                      src/sketching/seqsketchjaccard.rs:775,883 */
    KMU_FHASH_VALUE_MASKED = 1,   /* get_compressed_value() & mask   src/sketching/setsketchert.rs:1098-1104,
                                                                     src/aautils/setsketchert.rs:1235-1241 */
    KMU_FHASH_CANON_RAW = 2,      /* min(kmer, revcomp).0        src/base/kmercount.rs:313 */
    KMU_FHASH_CANON_INVHASH = 3,  /* intNN_hash(min(kmer,revcomp).0)  src/bin/datasketcher.rs:222-226,
                                                                      src/sketching/seqsketchjaccard.rs:931-935 */
    KMU_FHASH_INVHASH_RAW = 4,    /* intNN_hash(kmer.0)          src/sketching/seqminhash.rs:38,49 */
    KMU_FHASH_CANON_VALUE = 5,    /* min(kmer, revcomp).get_compressed_value()  src/base/kmercount.rs:313-316 */
    KMU_FHASH_CANON_NTHASH = 6,   /* canonical ntHash, 2-bit seed table  src/base/kmer.rs:76-95 (any k) */
    KMU_FHASH_CANON_NTHASH_8B = 7 /* canonical ntHash with the reference's ASCII table, bug-compatible:
                                     'G' and 'T' map to seed 0   src/base/nthash.rs:48-57,214-228 */
} kmu_fhash;

typedef enum kmu_algo {
    KMU_ALGO_PROB3A = 0, /* ProbMinHash3a  (probminhash crate), src/sketching/seqsketchjaccard.rs:211-260 */
    KMU_ALGO_SUPER = 1,  /* SuperMinHash   (probminhash crate), src/sketching/setsketchert.rs:255-297 */
    KMU_ALGO_SUPER2 = 2, /* SuperMinHash2  (integer sketch),   src/sketching/setsketchert.rs:963-1004.
                          * PROVISIONAL: the crate's fixed-point arithmetic is not in the reference tree; reference parity
                          * cannot be established in this build environment */
    KMU_ALGO_BOTTOMK = 3, /* MinHashCount / MinInvHashCountKmer, src/sketching/minhash.rs:62-99,219-265 */
    KMU_ALGO_PROB3 = 4,   /* ProbMinHash3 (SeqSketcher::sketch_probminhash3, seqsketchjaccard.rs:272-319): the same point
                           * process as ProbMinHash3a handled key by key; the signature is the same per-slot arg-min */
    /* PROVISIONAL (5, 6, 7): scheme from the cited papers, inner hash / variant choices of the `probminhash` crate
     * unknown here; reference parity impossible in this build environment */
    KMU_ALGO_OPTDENS = 5, /* OptDensHashSketch: one-permutation hashing + optimal densification (Shrivastava 2017),
                           * src/sketching/setsketchert.rs:343-463, src/aautils/setsketchert.rs:482-612; sig F32 / F64 */
    KMU_ALGO_REVOPTDENS = 6, /* RevOptDensHashSketch: the same with reverse densification (Mai et al. 2019), meant for
                           * sketches larger than the sequences, setsketchert.rs:474-599, aautils/setsketchert.rs:616-746 */
    KMU_ALGO_HLL = 7      /* HyperLogLogSketch: SetSketch registers (Ertl 2021), setsketchert.rs:640-896,
                           * aautils/setsketchert.rs:780-1011; sig U16 / U32 / U64; parameters: kmu_set_hll_params */
} kmu_algo;

typedef enum kmu_sig_type { KMU_SIG_U32 = 0, KMU_SIG_U64 = 1, KMU_SIG_F32 = 2, KMU_SIG_F64 = 3, KMU_SIG_U16 = 4 /* HLL only */ } kmu_sig_type;

/* std::hash::Hasher used to seed the per-element RNG */
typedef enum kmu_hasher {
    KMU_HASHER_NOHASH = 0, /* src/nohasher.rs:11-49: finish() = byte-swapped key on little-endian hosts */
    KMU_HASHER_FNV1A = 1,  /* fnv::FnvHasher, src/sketching/seqsketchjaccard.rs:346-349 */
    KMU_HASHER_INT64HASH = 2 /* BOTTOMK only: int64_hash(value as u64), MinInvHashCountKmer minhash.rs:223-233 */
} kmu_hasher;

typedef enum kmu_sketch_mode {
    KMU_MODE_PER_SEQ = 0, /* sketch_compressedkmer: one signature per sequence      setsketchert.rs:121-157 */
    KMU_MODE_ALL_SEQS = 1 /* sketch_compressedkmer_seqs: one signature for all       setsketchert.rs:160-202 */
} kmu_sketch_mode;

/* flags */
#define KMU_FLAG_RAND08 0x1u /* index draws as rand 0.8 `Uniform<usize>` (64-bit widening multiply) instead of
                                rand 0.9 (32-bit when the range fits u32); see DESIGN.md "unpinned" */

typedef struct kmu_device_cfg {
    int32_t device_id;     /* HIP device ordinal */
    int32_t async_device;  /* 1: KMU_MEM_DEVICE calls return without synchronising the stream */
    void *stream;          /* hipStream_t to enqueue on, or NULL to let the library create one */
    uint64_t workspace_bytes; /* 0 = grow on demand */
} kmu_device_cfg;

typedef struct kmu_sketch_params {
    int32_t algo;        /* kmu_algo */
    int32_t kmer_type;   /* kmu_kmer_type */
    int32_t kmer_size;
    int32_t sketch_size; /* m */
    int32_t sig_type;    /* kmu_sig_type; PROB3A/BOTTOMK: must match Kmer::Val width; SUPER / OPTDENS / REVOPTDENS: F32/F64 */
    int32_t hasher;      /* kmu_hasher */
    int32_t fhash;       /* kmu_fhash */
    int32_t block_size;  /* 0 = whole sequence; >0 = BlockSeqSketcher (src/sketching/seqblocksketch.rs:97-149) */
    int32_t mode;        /* kmu_sketch_mode */
    int32_t input_kind;  /* kmu_input_kind */
    int32_t mem;         /* kmu_mem */
    uint32_t flags;
} kmu_sketch_params;

typedef struct kmu_hash_params {
    int32_t kmer_type;
    int32_t kmer_size;
    int32_t fhash;
    int32_t input_kind;
    int32_t mem;
    uint32_t flags;
} kmu_hash_params;

typedef struct kmu_count_params {
    int32_t kmer_type;     /* DNA types only */
    int32_t kmer_size;
    int32_t counter_bits;  /* 8 or 16 (reference default 8: saturates at 255, src/base/kmercount.rs:1615) */
    int32_t flags;         /* 0, or KMU_COUNT_DISTRIBUTED [| KMU_COUNT_OWNER_HASH] (see "multi-GPU" below), KMU_COUNT_HINT_OCCURRENCES */
    uint64_t capacity_hint; /* expected number of distinct canonical k-mers (KMU_COUNT_HINT_OCCURRENCES: of k-mer occurrences) */
} kmu_count_params;

/* ---- context ------------------------------------------------------------------------------------- */
const char *kmu_version(void);
int kmu_device_count(void);
int kmu_create(const kmu_device_cfg *cfg, kmu_ctx **out);
void kmu_destroy(kmu_ctx *ctx);
const char *kmu_last_error(const kmu_ctx *ctx); /* ctx may be NULL: last error of a failed kmu_create */
int kmu_synchronize(kmu_ctx *ctx);
void *kmu_stream(kmu_ctx *ctx); /* the hipStream_t kernels are enqueued on */

/* Device buffers for callers that have no HIP binding of their own (a Rust or C host): allocate on the context's device,
 * copy in / out on the context's stream (both copies return when the data has arrived), then hand the pointers to the
 * KMU_MEM_DEVICE form of every entry point -- reads uploaded once can be ingested, sketched and counted without
 * crossing PCIe again.  kmu_dev_free waits for the context's stream first. */
int kmu_dev_alloc(kmu_ctx *ctx, uint64_t bytes, void **out);
int kmu_dev_free(kmu_ctx *ctx, void *p);
/* pinned host memory (hipHostMalloc): uploads from it and downloads into it are real DMA transfers that overlap kernels */
int kmu_host_alloc(kmu_ctx *ctx, uint64_t bytes, void **out);
int kmu_host_free(kmu_ctx *ctx, void *p);
int kmu_copy_to_device(kmu_ctx *ctx, void *dst_device, const void *src_host, uint64_t bytes);
int kmu_copy_to_host(kmu_ctx *ctx, void *dst_host, const void *src_device, uint64_t bytes);

/* SetSketchParams of the following KMU_ALGO_HLL calls on this context (probminhash::setsketcher::SetSketchParams, passed to
 * HyperLogLogSketch::new, setsketchert.rs:660-672): register base b > 1, rate a > 0, register limit q (values 0 .. q + 1).
 * m is the call's sketch_size.  Defaults: b = 1.001, a = 20, q = 65534. */
typedef struct kmu_hll_params {
    double b;
    double a;
    uint32_t q;
    uint32_t reserved;
} kmu_hll_params;
int kmu_set_hll_params(kmu_ctx *ctx, const kmu_hll_params *hp);

/* per-kernel device timing with hipEvents on the context stream (bench.py roofline object) */
int kmu_profile_enable(kmu_ctx *ctx, int on);
int kmu_profile_reset(kmu_ctx *ctx);
/* returns number of kernels known; fills up to `cap` entries */
typedef struct kmu_kernel_stat {
    char name[48];
    uint64_t launches;
    double total_ms;
} kmu_kernel_stat;
int kmu_profile_get(kmu_ctx *ctx, kmu_kernel_stat *stats, int cap);

/* ---- L0: alphabet + Sequence::new(raw, 2)  (src/base/alphabet.rs:119-127,162-168; sequence.rs:25-106) ---- */
/* number of bytes b with !is_acgt(b) per sequence (alphabet.rs:28-31); counts_out[n_seq] */
int kmu_count_non_acgt(kmu_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int mem,
                       uint64_t *counts_out);
/* pack every sequence as Sequence::new(raw,2).  packed_offsets_out[n_seq+1] (bytes) is written first;
 * packed_out must hold sum(ceil(L_i/4)) bytes.  Returns KMU_E_NON_ACGT if any byte is not in ACGTacgt. */
int kmu_pack2b(kmu_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int mem,
               uint8_t *packed_out, uint64_t *packed_offsets_out);

/* ---- L1/L2: k-mer generation + canonicalise + hash, one value per k-mer start position -------------
 * KmerSeqIterator::next (src/base/kmergenerator.rs:75-106) followed by the fhash closure.
 * out[offsets[i] + p] (uint64, zero-extended for u32 types) for p in [0, L_i-k+1); other entries (the last k-1 positions
 * of a sequence) untouched in KMU_MEM_DEVICE arrays, zero in KMU_MEM_HOST arrays (the array comes back whole).
 * For KMU_INPUT_PACKED2 `packed_offsets` gives the byte offset of every sequence (may be NULL for ASCII). */
int kmu_kmer_hashes(kmu_ctx *ctx, const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                    const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *out);

/* KmerGenerationPattern::generate_kmer_pattern_in_range (src/base/kmergenerator.rs:126; impls :271-303,368-408,491-526;
 * amino acids src/aautils/kmeraa.rs:780-812,870-899): the k-mers that lie inside bases [range_begin[i], range_end[i]) of
 * sequence i -- KmerSeqIterator::set_range (kmergenerator.rs:66-68) on IterSequence::set_range (src/base/sequence.rs:562-585).
 * out[offsets[i] + p] for p in [range_begin[i], range_end[i] - k]; other entries untouched.  A range with end <= begin or
 * end > L_i is the reference's `Err(())` (unwrapped by its callers: a panic): KMU_E_BAD_ARG.  (The reference's debug build
 * also underflows for end < 4, sequence.rs:574; a release build wraps back to the intended byte: ends 1..3 are served.) */
int kmu_kmer_hashes_range(kmu_ctx *ctx, const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                          const uint64_t *packed_offsets, uint32_t n_seq, const uint64_t *range_begin,
                          const uint64_t *range_end, uint64_t *out);

/* KmerGenerationPattern::generate_kmer_distribution / KmerGenerator::generate_weighted_kmer (kmergenerator.rs:130,176; impls
 * :245-269,339-366,447-489; amino acids kmeraa.rs:752-778,845-868): the DISTINCT values fhash(kmer) of every sequence with
 * their multiplicities (u32 like the reference's FnvHashMap<T, u32>).  The pairs of sequence i are
 * kmers_out / mult_out [dist_offsets_out[i] .. dist_offsets_out[i + 1]), in no particular order (a hash map upstream).
 * Call with kmers_out == NULL for the sizes: *n_out = number of pairs, dist_offsets_out (n_seq + 1 entries, may be NULL).
 * KMU_FHASH_IDENTITY_RAW gives the k-mers themselves (`.0`), what the reference's map is keyed by. */
int kmu_kmer_distribution(kmu_ctx *ctx, const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                          const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *kmers_out, uint32_t *mult_out,
                          uint64_t cap, uint64_t *dist_offsets_out, uint64_t *n_out);

/* ntHash per k-mer position (src/base/nthash.rs; NtHash for 2-bit k-mers src/base/kmer.rs:45-145).
 *   table KMU_NTHASH_TABLE_2B: seeds of BASE_MAPPING_2B (nthash.rs:28-30) -- Kmer32bit / Kmer16b32bit::nthash_init,
 *                              nthash_canonical_init (kmer.rs:48-95); ASCII or packed input
 *         KMU_NTHASH_TABLE_8B: the reference's ASCII table BASE_MAPPING_8B bug for bug ('G' and 'T' hit zero entries,
 *                              nthash.rs:48-57) -- nthash_init_8b :153-161, nthash_rcomp_init_8b :182-189,
 *                              nthash_canonical_init_8b :214-228; upper-case ASCII input only
 *   mode  KMU_NTHASH_CANONICAL: min(forward, rcomp) and its strand (0 forward, 1 reverse; forward on ties, :223-227)
 *         KMU_NTHASH_FORWARD / KMU_NTHASH_RCOMP: one strand only (strand byte 0 / 1)
 *   n_hashes > 1: from_one_hash_val_to_mult_hash (:63-72) -- nthash_mult_canonical_init_8b :255-269, kmer.rs:123-131.
 * hashes_out[(offsets[i] + p) * n_hashes + j], strand_out[offsets[i] + p] (may be NULL) for p in [0, L_i - k + 1).
 * The value at every position is the *_init value of the k-mer starting there = what a correct roll reaches (the property
 * the reference tests, nthash.rs:333,379).  Not reproduced: the state resets that make the reference's own
 * nthash_mult_canonical_cycle_8b (:279-280) and the 2-bit nthash_canonical_cycle (kmer.rs:98-99) differ from their init forms.
 * 1 <= kmer_size <= 32. */
typedef enum kmu_nthash_mode { KMU_NTHASH_CANONICAL = 0, KMU_NTHASH_FORWARD = 1, KMU_NTHASH_RCOMP = 2 } kmu_nthash_mode;
typedef enum kmu_nthash_table { KMU_NTHASH_TABLE_2B = 0, KMU_NTHASH_TABLE_8B = 1 } kmu_nthash_table;
typedef struct kmu_nthash_params {
    int32_t kmer_size;
    int32_t table;      /* kmu_nthash_table */
    int32_t mode;       /* kmu_nthash_mode */
    int32_t n_hashes;   /* >= 1 */
    int32_t input_kind; /* kmu_input_kind */
    int32_t mem;        /* kmu_mem */
} kmu_nthash_params;
int kmu_nthash(kmu_ctx *ctx, const kmu_nthash_params *p, const uint8_t *bases, const uint64_t *offsets,
               const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *hashes_out, uint8_t *strand_out);

/* The same values without the gaps: fhash of every k-mer of sequence 0, then of sequence 1, ... (uint64, zero-extended).
 * Call with out == NULL for the number of values in *n_out.  (What a rank feeds to the owner exchange of a distributed
 * ProbMinHash sketch, see kmu_sketch_hashed_partial.) */
int kmu_kmer_hashes_compact(kmu_ctx *ctx, const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                            const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *out, uint64_t cap, uint64_t *n_out);

/* ---- L3 sketching: SeqSketcherT::sketch_compressedkmer{,_seqs}  (src/sketching/setsketchert.rs:54-80),
 * SeqSketcher::sketch_probminhash3a / sketch_superminhash (src/sketching/seqsketchjaccard.rs:211,328),
 * BlockSeqSketcher::blocksketch_sequences (src/sketching/seqblocksketch.rs:152-167),
 * SeqSketcherAAT (src/aautils/setsketchert.rs:42-72) ------------------------------------------------
 * sig_out: PER_SEQ, block_size == 0 : n_seq rows of sketch_size signatures, row i <-> sequence i
 *          PER_SEQ, block_size  > 0 : rows block_row_offsets[i] .. block_row_offsets[i+1], one row per block
 *                                     (block_row_offsets from kmu_block_layout; may be NULL if block_size==0)
 *          ALL_SEQS                 : 1 row
 * BOTTOMK: sig_out rows are sketch_size uint64 hashes ascending, padded with UINT64_MAX; counts_out (may be
 *          NULL) gets sketch_size uint32 multiplicities per row. */
int kmu_sketch(kmu_ctx *ctx, const kmu_sketch_params *p, const uint8_t *bases, const uint64_t *offsets,
               const uint64_t *packed_offsets, uint32_t n_seq, const uint64_t *block_row_offsets, void *sig_out,
               uint32_t *counts_out);
/* nb_blocks_i = ceil(L_i / block_size) (seqblocksketch.rs:108-112); block_row_offsets_out[n_seq+1], host memory,
 * offsets in host memory */
int kmu_block_layout(const uint64_t *offsets, uint32_t n_seq, uint32_t block_size, uint64_t *block_row_offsets_out);

/* The reads ONCE, both results: the per-sequence signatures of kmu_sketch and the k-mer counts of kmu_count_add_reads for
 * one batch of unpacked reads -- the reference's alternation "read a pack of sequences, sketch it" (src/bin/datasketcher.rs:
 * 243-260) and its counting pass (src/bin/parsefastq.rs:215-236) as one stream-ordered pipeline.
 *   p->mem == KMU_MEM_HOST   (bases, offsets, sig_out in host memory, best pinned: kmu_host_alloc): the bases cross PCIe
 *       once, in chunks of whole reads on an upload stream; the sketch kernels of chunk i run under the upload of chunk
 *       i + 1, the signature rows of chunk i go down on a download stream under the kernels that follow, and the count
 *       build runs on the resident reads under the last downloads.  Returns when everything is in host memory.
 *   p->mem == KMU_MEM_DEVICE: both results from the resident reads; a distributed counter's all-to-all is in flight
 *       while the reads are sketched.
 * counter may be NULL (signatures only).  Whole sequences, one signature per sequence (mode PER_SEQ, block_size 0), ASCII.
 * KMU_PIPE_CHUNK_MB sets the chunk size of the host form (default 512). */
int kmu_sketch_count(kmu_ctx *ctx, const kmu_sketch_params *p, kmu_counter *counter, const uint8_t *bases,
                     const uint64_t *offsets, uint32_t n_seq, void *sig_out);

/* fallback for arbitrary closures: the host evaluates fhash, the device does multiset + sketch only.
 * hashed[offsets[i] .. offsets[i+1]) are the hashed k-mers of sequence i (uint32 or uint64 per p->sig_type /
 * kmer_type width). */
int kmu_sketch_hashed(kmu_ctx *ctx, const kmu_sketch_params *p, const void *hashed, const uint64_t *offsets,
                      uint32_t n_seq, void *sig_out, uint32_t *counts_out);

/* One signature for sequences spread over several GPUs (sketch_compressedkmer_seqs across ranks, one process per GPU):
 * every rank turns its share into per-slot minima -- a "partial" of kmu_sketch_partial_words(p) uint64 words -- the ranks
 * exchange these small arrays (all-gather over RCCL) and every rank merges them into the signature.
 *   SuperMinHash / SuperMinHash2 / OptDens / RevOptDens treat k-mer occurrences independently: a rank's share is its reads.
 *   ProbMinHash weighs a key by its multiplicity over ALL sequences: the shares must be DISJOINT KEY SETS -- hash the k-mers
 *   (kmu_kmer_hashes), exchange the values by owner, and feed what a rank receives to kmu_sketch_hashed_partial.
 * `mode` and `block_size` of the parameters are ignored (one multiset over everything given); `mem` says where the partial
 * and the other buffers live.  A partial of no data at all merges as the neutral element. */
uint32_t kmu_sketch_partial_words(const kmu_sketch_params *p);
int kmu_sketch_partial(kmu_ctx *ctx, const kmu_sketch_params *p, const uint8_t *bases, const uint64_t *offsets,
                       const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *partial_out);
int kmu_sketch_hashed_partial(kmu_ctx *ctx, const kmu_sketch_params *p, const void *hashed, const uint64_t *offsets, uint32_t n_seq,
                              uint64_t *partial_out);
/* partials: n_parts x kmu_sketch_partial_words(p) words; sig_out: sketch_size signatures */
int kmu_sketch_merge_partials(kmu_ctx *ctx, const kmu_sketch_params *p, const uint64_t *partials, uint32_t n_parts, void *sig_out);

/* ---- L3 counting: KmerCountT (src/base/kmercount.rs:48-59), KmerCounter::insert_kmer :241-267,
 * count_kmer_threaded_one_to_many :881-974, KmerCounterPool :424-565 --------------------------------- */
/* KMU_COUNT_HINT_OCCURRENCES (kmu_count_params.flags): capacity_hint counts the k-mer OCCURRENCES the caller is going to add (what
 * a caller knows: the bases of its reads), not distinct k-mers.  The table is then allocated by the first add and sized from the
 * duplication that add measures on a key sample of its batch (one extra pass over those reads, once per counter): occurrences /
 * ratio / 0.70 slots.  The reference's filters are sized the same blind way -- capacity 3e9 / n whatever the reads,
 * src/base/kmercount.rs:888-892 --; a table nine tenths empty costs nothing there, here every partitioned build writes the image.
 * Batches added later must fit the table the first one made (KMU_E_TABLE_FULL otherwise): give a distinct-k-mer hint where the
 * first batch is not representative.  Until that add kmu_count_table_info reports nslots = 0. */
#define KMU_COUNT_HINT_OCCURRENCES 0x4
int kmu_count_create(kmu_ctx *ctx, const kmu_count_params *p, kmu_counter **out);
void kmu_count_destroy(kmu_counter *c);
int kmu_count_reset(kmu_counter *c);
/* insert every canonical k-mer (min(kmer, revcomp), kmercount.rs:938) of every sequence */
int kmu_count_add_reads(kmu_counter *c, const uint8_t *bases, const uint64_t *offsets,
                        const uint64_t *packed_offsets, uint32_t n_seq, int input_kind, int mem);
/* insert pre-computed canonical compressed values (KmerCountT::insert_kmer, one call per value) */
int kmu_count_add_kmers(kmu_counter *c, const uint64_t *canon_kmers, uint64_t n, int mem);
/* KmerCountT::get_count (kmercount.rs:270-277): 0 never seen, 1 singleton, else min(multiplicity, 2^bits-1) */
int kmu_count_query(kmu_counter *c, const uint64_t *canon_kmers, uint64_t n, int mem, uint32_t *counts_out);
int kmu_count_nb_distinct(kmu_counter *c, uint64_t *out); /* kmercount.rs:280-282 */
int kmu_count_nb_unique(kmu_counter *c, uint64_t *out);   /* kmercount.rs:285-287 */
/* sum of the (unsaturated) multiplicities held = k-mer occurrences inserted: the conservation check of a build */
int kmu_count_nb_occurrences(kmu_counter *c, uint64_t *out);
/* k-mers whose in-table count has reached kmu_count_table_info_t.count_ceiling (a homopolymer in a deep read set): for them the sum
 * above and a raw export are lower bounds -- an inexact total then reads "saturated", not "k-mers lost" */
int kmu_count_nb_saturated(kmu_counter *c, uint64_t *out);
/* The table in HBM (no counterpart upstream: the reference's filters size themselves, kmercount.rs:70-123).  Slots: whole regions
 * of 4096 -- capacity_hint / 0.70 of them, NOT rounded to a power of two (round 5: the image of a big table is written by every
 * partitioned build).  Big tables (more than 2048 regions whose index is worth >= 11 bits with 8-bit counters, >= 17 with 16-bit
 * ones) keep ONE 8-byte word per slot -- the bits of the table hash that the slot's position does not already say, and a count field of
 * count_field_bits that stops a little below its maximum (every reader saturates at 2^counter_bits - 1 like
 * kmercount.rs:1615; kmu_count_nb_occurrences is exact while no k-mer occurs 2^count_field_bits - 1024 times) --, small ones
 * a 64-bit key + a 32-bit count. */
typedef struct kmu_count_table_info_t {
    uint64_t nslots;
    uint64_t table_bytes;
    uint32_t bytes_per_slot;   /* 8 or 12 */
    uint32_t count_field_bits; /* width of the in-table count */
    uint64_t count_ceiling;    /* the largest count a slot holds: 2^32 - 1 (12-byte slots), 2^count_field_bits - 1024 (8-byte slots).  Queries saturate at 2^counter_bits - 1 long before;
                                  kmu_count_nb_occurrences and the raw counts of kmu_count_export_part are exact while
                                  kmu_count_nb_saturated is 0 */
} kmu_count_table_info_t;
int kmu_count_table_info(const kmu_counter *c, kmu_count_table_info_t *out);
/* dump of (canonical k-mer, count) for count >= min_count (dump_kmer_counter, kmercount.rs:500-525 uses 2).
 * Call with kmers_out == NULL to get the number of records in *n_out; records are sorted by k-mer value. */
int kmu_count_dump(kmu_counter *c, uint32_t min_count, uint64_t *kmers_out, uint32_t *counts_out, uint64_t cap,
                   uint64_t *n_out);
/* KmerCounter::eliminate_once_kmer (kmercount.rs:110-117): forget the k-mers seen once; counts >= 2 are kept */
int kmu_count_eliminate_once(kmu_counter *c);
/* The k-mers seen exactly once, with where they sit (KmerFilter1::dump_in_file_once_kmer16b32bit, kmercount.rs:1031-1082;
 * the `Unicity` branch of parsefastq): for every k-mer occurrence of the given reads, in (sequence, position) order, whose
 * canonical k-mer has count exactly 1 in the counter: canonical value, sequence number, k-mer rank in its sequence.
 * Unpacked (ASCII) input.  Call with kmers_out == NULL for the number of records in *n_out. */
int kmu_count_once_positions(kmu_counter *c, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int mem,
                             uint64_t *kmers_out, uint32_t *numseq_out, uint32_t *numkmer_out, uint64_t cap, uint64_t *n_out);
/* multi-GPU merge (one process per GPU): export the entries whose owner (int64_hash(kmer) % n_parts,
 * kmercount.rs:412-420) is `part` as device/host arrays, and merge entries received from peers.
 * The exchange itself is done by the host layer (RCCL all_to_all over xGMI). */
int kmu_count_export_part(kmu_counter *c, uint32_t part, uint32_t n_parts, uint64_t *kmers_out, uint32_t *counts_out,
                          uint64_t cap, int mem, uint64_t *n_out);
int kmu_count_merge_entries(kmu_counter *c, const uint64_t *kmers, const uint32_t *counts, uint64_t n, int mem);
/* multi-GPU counting, the throughput path: the canonical k-mers of the reads grouped by owner rank
 * (`int64_hash(kmer) % n_parts`, the reference's own key-space dispatch, kmercount.rs:412-420,942).
 * *dev_kmers_out points to a device buffer owned by the counter's context (valid until its next call);
 * part_bounds_out[n_parts + 1] (host) delimits the group of every owner.  The caller exchanges the groups
 * (all-to-all over RCCL) and feeds what it receives to kmu_count_add_kmers. */
int kmu_count_extract_by_owner(kmu_counter *c, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int mem,
                                uint32_t n_parts, uint64_t **dev_kmers_out, uint64_t *part_bounds_out);
/* drop every entry not owned by `part` (after the exchange each rank keeps only its key range) */
int kmu_count_retain_part(kmu_counter *c, uint32_t part, uint32_t n_parts);

/* ---- multi-GPU: one rank per GPU, RCCL over xGMI ------------------------------------------------------------------
 * The reference's parallel drivers are threads of one process: count_kmer_threaded_one_to_many (src/base/kmercount.rs:881-974)
 * dispatches every canonical k-mer to the thread that owns it (`int64_hash(kmer) % n`, :412-420, :942) over channels, and
 * KmerCounterPool (:424-565) keeps one counter per thread; sketching is a rayon map over reads
 * (src/sketching/seqsketchjaccard.rs:245-248).  Here a rank is a GPU with its own kmu_ctx (one process per GPU, or one
 * thread per GPU), the channel is ONE all-to-all over xGMI inside the library, and sketching shards reads with no
 * collective.  Rank 0 creates an id and hands its 128 bytes to the other ranks by any means the host has (a file, MPI, a
 * socket, torch.distributed); every rank then calls kmu_comm_init (collective).  RCCL is bound when the first of these
 * calls is made, to the copy already mapped into the process if there is one. */
#define KMU_COMM_ID_BYTES 128
typedef struct kmu_comm_id { char bytes[KMU_COMM_ID_BYTES]; } kmu_comm_id; /* an ncclUniqueId */
int kmu_comm_get_id(kmu_comm_id *out);
int kmu_comm_init(kmu_ctx *ctx, const kmu_comm_id *id, int rank, int nranks);
/* A transport supplied by the host instead (a host that owns a communicator already; tests with more ranks than GPUs).
 * alltoallv (may be NULL: the library's COPY transport, below): device pointers; counts and displacements per peer in elements of elem_bytes; the library has synchronised
 * `stream` before the call and expects the received data to be visible to the device at return.  allgather: host memory,
 * `bytes` from every rank, rank order.  Both return 0 on success. */
typedef int (*kmu_alltoallv_fn)(void *user, const void *send_dev, const uint64_t *send_counts, const uint64_t *send_displs,
                                void *recv_dev, const uint64_t *recv_counts, const uint64_t *recv_displs, uint32_t elem_bytes,
                                void *stream);
typedef int (*kmu_allgather_fn)(void *user, const void *send_host, void *recv_host, uint64_t bytes);
int kmu_comm_init_custom(kmu_ctx *ctx, int rank, int nranks, kmu_alltoallv_fn alltoallv, kmu_allgather_fn allgather, void *user);
/* The COPY transport (round 5): the all-to-all of a distributed add as N - 1 device-to-device copies straight into the peers' receive
 * buffers -- mapped through HIP IPC (hipIpcGetMemHandle / hipIpcOpenMemHandle) when the peer is another process of this node, used as
 * they are when it is a thread of this one -- on the communicator's exchange stream.  No kernel of a collective library has to find
 * a CU under the persistent sketch kernels a step runs the exchange under (RCCL's send / receive kernels do: measured on one GPU,
 * profiles/r05_reserve.txt).  The communicator's all-gather (RCCL's, or the host's) carries the handles and closes the exchange with
 * a host barrier.  One node only.  kmu_comm_set_transport(ctx, KMU_TRANSPORT_COPY) switches an existing communicator over (every
 * rank alike, between exchanges); kmu_comm_init_custom with alltoallv == NULL creates one that has only this transport.  An exchange
 * for which some rank cannot export its receive buffer goes -- on every rank: all read the same published rows -- through the
 * communicator's other data path (RCCL, or the host's all-to-all) if it has one, and fails with KMU_E_RCCL otherwise. */
typedef enum kmu_transport { KMU_TRANSPORT_DEFAULT = 0 /* RCCL, or the host's all-to-all function */, KMU_TRANSPORT_COPY = 1 } kmu_transport;
int kmu_comm_set_transport(kmu_ctx *ctx, int transport);
int kmu_comm_transport(const kmu_ctx *ctx); /* kmu_transport of the context's communicator */
int kmu_comm_destroy(kmu_ctx *ctx); /* also done by kmu_destroy */
int kmu_comm_rank(const kmu_ctx *ctx);   /* -1 without a communicator */
int kmu_comm_nranks(const kmu_ctx *ctx); /* 0 without a communicator */
int kmu_comm_allgather(kmu_ctx *ctx, const void *send_host, void *recv_host, uint64_t bytes); /* host memory, collective */

/* Distributed counting.  A counter created with KMU_COUNT_DISTRIBUTED on a context that has a communicator is one member of
 * a KmerCounterPool spread over the ranks: kmu_count_add_reads takes THIS rank's reads (collective: every rank calls it,
 * with its own shard, possibly empty), kmu_count_finalize (collective) completes the exchange; afterwards rank r's counter
 * holds exactly the canonical k-mers with owner r with their multiplicities over ALL ranks' reads, and every query / dump /
 * statistic of this header sees that partition.
 * Who owns a k-mer (kmu_count_owner_kind; every rank of a pool must create its counter alike):
 *   KMU_OWNER_MINIMIZER  (default where a minimizer window fits: Kmer64bit, 17 <= k <= 31) the owner is a function of the k-mer's
 *                MINIMIZER -- the canonical m-mer of smallest hash among the k - m + 1 it contains (m = 9 .. 15 by k), the same on
 *                both strands: kmu_kmer_owner_minimizer.  Consecutive k-mers of a read mostly share it, so the occurrences travel
 *                as SUPER-K-MER RECORDS: a run of up to 16 consecutive k-mers with one owner as its 2-bit bases, 12 bytes, ~1.35 B
 *                per k-mer at k = 31 instead of one 8-byte message per k-mer (the reference's unit of dispatch, kmercount.rs:936-943).
 *   KMU_OWNER_HASH       (KMU_COUNT_OWNER_HASH, KMU_COUNT_OWNER=hash, and every k <= 16) the reference's own dispatch
 *                kmu_kmer_owner: int32_hash / int64_hash(kmer) % n, DispatchableT (kmercount.rs:382-420).
 * Routes, chosen per add from the measured duplication (occurrences / distinct k-mers, estimated on a hash sample of this batch
 * over all ranks) with a cost model (kmu_comm_stats; KMU_COUNT_ROUTE=occurrences|superkmers|merge forces one):
 *   OCCURRENCES  (hash owners) every k-mer occurrence travels to its owner (8 B each), the owner builds its table from what it
 *                receives: the reference's own dispatch.
 *   SUPERKMERS   (minimizer owners) the same with 12-byte records in place of k-mers; the owner expands them inside the first
 *                level of its partitioned build.
 *   MERGE        every rank counts its shard locally; at finalize the (k-mer, count) entries it does not own travel
 *                (12 B per DISTINCT k-mer) and are added to the owner's table: less traffic only once a k-mer occurs many times
 *                per rank (more than ~1.5 with hash owners, ~9 with minimizer owners), at the price of building twice. */
#define KMU_COUNT_DISTRIBUTED 0x1 /* kmu_count_params.flags */
#define KMU_COUNT_OWNER_HASH 0x2  /* with KMU_COUNT_DISTRIBUTED: owners by the reference's dispatch (kmu_kmer_owner) */
typedef enum kmu_owner_kind { KMU_OWNER_HASH = 0, KMU_OWNER_MINIMIZER = 1 } kmu_owner_kind;
typedef enum kmu_count_route { KMU_ROUTE_NONE = 0, KMU_ROUTE_OCCURRENCES = 1, KMU_ROUTE_MERGE = 2, KMU_ROUTE_SUPERKMERS = 3 } kmu_count_route;
typedef struct kmu_comm_stats {
    int32_t route;                 /* kmu_count_route of the last distributed kmu_count_add_reads */
    int32_t sample_shift;          /* the duplication was measured on the k-mers whose owner hash has this many zero bits */
    double dup_ratio;              /* occurrences / distinct over all ranks (sampled); 0 = not measured */
    uint64_t kmers_local;          /* k-mer occurrences of this rank's shard (last add) */
    uint64_t bytes_occurrences;    /* what OCCURRENCES / SUPERKMERS moves off this rank for that add: 8 B x the occurrences (12 B x
                                      the records) that other ranks own */
    uint64_t bytes_merge;          /* what MERGE moves off this rank: 12 B x distinct x (N-1)/N (from dup_ratio until
                                      finalize has run, exact afterwards) */
    uint64_t bytes_sent;           /* bytes that really left this rank / arrived, last add + its finalize */
    uint64_t bytes_received;
    double model_ms_occurrences;   /* the cost model's estimates the choice was made from (OCCURRENCES or SUPERKMERS; MERGE) */
    double model_ms_merge;
    int32_t owner_kind;            /* kmu_owner_kind of the counter of the last add */
    int32_t exchanges;             /* all-to-alls of the last add + its finalize */
    uint64_t records_local;        /* super-k-mer records of this rank's shard (0 with hash owners) */
    double exchange_ms;            /* MEASURED duration of those all-to-alls: events around them on the exchange stream (RCCL), wall
                                      time of the host's function (kmu_comm_init_custom).  An exchange still in flight is counted by the next query. */
    double exchange_gbps_out;      /* bytes_sent / exchange_ms and bytes_received / exchange_ms, in GB/s: the link rate the route */
    double exchange_gbps_in;       /* model assumes as KMU_XGMI_GBPS */
} kmu_comm_stats;
int kmu_comm_get_stats(const kmu_ctx *ctx, kmu_comm_stats *out);
int kmu_count_finalize(kmu_counter *c);
int kmu_count_owner_kind(const kmu_counter *c); /* kmu_owner_kind of a distributed counter (KMU_OWNER_HASH for any other) */
/* owner of canonical k-mer values in an n_parts-way pool: int32_hash / int64_hash(kmer) % n_parts by the width of the
 * k-mer type (Kmer32bit / Kmer16b32bit: kmercount.rs:386-402; Kmer64bit :412-420).  Host arithmetic, no device needed. */
int kmu_kmer_owner(int kmer_type, const uint64_t *canon_kmers, uint64_t n, uint32_t n_parts, uint32_t *owners_out);
/* the minimizer owner of Kmer64bit values of kmer_size bases (17 .. 31; forward or canonical value: both strands agree).  Host
 * arithmetic, no device needed. */
int kmu_kmer_owner_minimizer(int kmer_size, const uint64_t *canon_kmers, uint64_t n, uint32_t n_parts, uint32_t *owners_out);
/* The two halves of the SUPERKMERS route for hosts that run the exchange themselves (kmu_count_extract_by_owner's counterpart).
 * A record is 12 bytes = three uint32: up to 46 bases, 2 bits each, the first in bits 31..30 of word 0, zeros behind the last;
 * bits 3..0 of word 2 = number of k-mers - 1 (1 .. 16 k-mers: consecutive k-mers of one read with one minimizer owner).
 * kmu_count_extract_superkmers: the records of the reads grouped by owner; *dev_records_out points to a device buffer owned
 * by the counter's context (valid until its next call), record_bounds_out[n_parts + 1] (host, in records) delimits the group of
 * every owner, kmers_per_part_out[n_parts] (host, may be NULL) says how many k-mers each group holds.
 * kmu_count_add_superkmers: the k-mers of n_records records into the table (KMU_MEM_DEVICE or KMU_MEM_HOST). */
int kmu_count_extract_superkmers(kmu_counter *c, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int mem,
                                 uint32_t n_parts, void **dev_records_out, uint64_t *record_bounds_out, uint64_t *kmers_per_part_out);
int kmu_count_add_superkmers(kmu_counter *c, const void *records, uint64_t n_records, int mem);

/* ---- L-1 ingest: the step before the path ----------------------------------------------------
 * FASTQ / FASTA text -> the accepted reads as (bases, offsets), in file order.  kmu_ingest_fastq: 4-line records,
 * "\n" or "\r\n".
 * Rule of the reference's readers: a record with any byte outside ACGTacgt is dropped and counted
 * (readblockseq, src/bin/datasketcher.rs:358-388; parse_with_needletail, src/io.rs:37-57).
 * Call with bases_out == offsets_out == NULL to get the sizes in *info, then with buffers of at least
 * info->kept_bases bytes and info->n_kept + 1 offsets.  record_index_out (optional, n_kept entries) = position of every
 * kept read among the records of the text.  `text` must be 16-byte aligned in KMU_MEM_DEVICE mode.
 * Errors: KMU_E_BAD_ARG for a malformed or truncated record (the reference: `record.expect("invalid record")`). */
typedef struct kmu_ingest_info {
    uint64_t n_records;    /* records in the text */
    uint64_t n_kept;       /* records made of ACGTacgt only */
    uint64_t kept_bases;   /* sum of their lengths */
    uint64_t n_bases;      /* bases of all records        (io.rs:40) */
    uint64_t nb_bad_bases; /* non-ACGT bytes in sequences (io.rs:42) */
    uint64_t nb_bad_reads; /* dropped records             (io.rs:47, datasketcher.rs:368) */
} kmu_ingest_info;
int kmu_ingest_fastq(kmu_ctx *ctx, const uint8_t *text, uint64_t n_bytes, int mem, uint8_t *bases_out, uint64_t bases_cap,
                     uint64_t *offsets_out, uint64_t offsets_cap, uint32_t *record_index_out, kmu_ingest_info *info);

/* FASTA, needletail's other format (what gsearch feeds): a line that starts with '>' opens a record; its sequence is the
 * following lines up to the next such line with the line ends removed (record.seq()).  Same rule, outputs and calling
 * convention as kmu_ingest_fastq.  KMU_E_BAD_ARG if the text does not start with '>'. */
int kmu_ingest_fasta(kmu_ctx *ctx, const uint8_t *text, uint64_t n_bytes, int mem, uint8_t *bases_out, uint64_t bases_cap,
                     uint64_t *offsets_out, uint64_t offsets_cap, uint32_t *record_index_out, kmu_ingest_info *info);
/* needletail::parse_fastx_file (src/io.rs:37, src/bin/datasketcher.rs:211): the first byte names the format,
 * '>' FASTA, '@' FASTQ; anything else is KMU_E_BAD_ARG.  (Compressed input is not handled: decompress first.) */
int kmu_ingest_fastx(kmu_ctx *ctx, const uint8_t *text, uint64_t n_bytes, int mem, uint8_t *bases_out, uint64_t bases_cap,
                     uint64_t *offsets_out, uint64_t offsets_cap, uint32_t *record_index_out, kmu_ingest_info *info);

/* ---- L4 signature comparison: what the callers do with the signatures next --------------------
 * Rows are compared as raw words (4 bytes for KMU_SIG_U32 / F32, 8 for U64 / F64).
 * out[p] = number of slots t with A[ia[p]][t] == B[ib[p]][t]:
 *   probminhash_get_jaccard_objects (src/sketching/seqsketchjaccard.rs:86-108): jaccard = out / m
 *   DistBlockSketched::eval / distance_jaccard_serial (src/sketching/seqblocksketch.rs:419-440): (m - out) / m, and 1.0
 *   when both blocks belong to the same sequence (the caller's rule); DistHamming of datasketcher.rs:156-185 likewise.
 * sig_a has na rows of m words, sig_b nb rows (they may be the same array). */
int kmu_sig_equal_pairs(kmu_ctx *ctx, const void *sig_a, uint32_t na, const void *sig_b, uint32_t nb, uint32_t m,
                        int sig_type, const uint32_t *ia, const uint32_t *ib, uint64_t n_pairs, int mem, uint32_t *out);
/* all pairs: out[i * nb + j] (m <= 65535) */
int kmu_sig_equal_matrix(kmu_ctx *ctx, const void *sig_a, uint32_t na, const void *sig_b, uint32_t nb, uint32_t m,
                         int sig_type, int mem, uint16_t *out);
/* minhash_distance / mininvhash_distance (src/sketching/minhash.rs:134-190, :295-340) on KMU_ALGO_BOTTOMK rows
 * (ascending hashes, u64::MAX padding).  out[3p..3p+2] = common, total, i: jaccard = common / total,
 * containment = common / i (MinHashDist). */
int kmu_minhash_distance_pairs(kmu_ctx *ctx, const uint64_t *hashes_a, uint32_t na, const uint64_t *hashes_b, uint32_t nb,
                               uint32_t m, const uint32_t *ia, const uint32_t *ib, uint64_t n_pairs, int mem,
                               uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* KMU_H */
