/*
 * kmu_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-thread restatement of the reference algorithms on the hot path of
 * jean-pierreBoth/kmerutils, each function citing the reference file:line it follows.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / reported baseline.  The product path (kmerutils_amd/csrc, libkmu.so) never links or calls it.
 *
 * PARITY STATUS
 *   pinned   : alphabet / packing / k-mer values / reverse complement / iteration / ntHash / multiset /
 *              counting semantics / AA k-mers -- restated from source in the reference tree and checked against
 *              every known-answer test the reference holds for them (tests/test_oracle_kat.py).
 *              Restated from source in the tree, without a reference KAT of their own (hand cases in
 *              tests/test_oracle_kat.py): the FASTQ reader rule (io.rs / datasketcher.rs), the signature comparison
 *              functions (seqsketchjaccard.rs, seqblocksketch.rs, minhash.rs).
 *   UNPINNED : Wang invertible hashes, xoshiro seeding, ProbMinHash3a / ProbMinHash3, SuperMinHash, SuperMinHash2,
 *              OptDensMinHash / RevOptDensMinHash, SetSketcher internals.
 *              They live in the un-vendored crate `probminhash = "0.1"` (reference Cargo.toml:89), which is
 *              not in /root/reference and cannot be built here (no Rust toolchain).  They are restated from
 *              Ertl's papers (arXiv 1706.05698, 1911.00675), Shrivastava (ICML 2017) and Mai et al. (UAI 2019) for the
 *              densified sketches, and the crate's structure as recalled; the
 *              reference's own tests pin them only statistically => "parity unpinned".
 */
#ifndef KMU_ORACLE_H
#define KMU_ORACLE_H

#include "../include/kmu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- L0 ---- */
int kmo_encode2b(uint8_t c);   /* Alphabet2b::encode; -1 where the reference panics */
uint8_t kmo_decode2b(uint8_t code);
uint64_t kmo_count_non_acgt(const uint8_t *raw, uint64_t n);
/* Sequence::new(raw, 2): returns number of bytes written, or -1 on a non-ACGT byte */
int64_t kmo_pack2b(const uint8_t *raw, uint64_t n, uint8_t *out);
/* Sequence::encode_and_add semantics: invalid bytes are skipped. returns number of bases kept */
int64_t kmo_pack2b_filtered(const uint8_t *raw, uint64_t n, uint8_t *out);
uint8_t kmo_get_base(const uint8_t *packed, uint64_t pos);
int kmo_encode_aa(uint8_t c); /* aautils Alphabet::encode; -1 where the reference panics */

/* ---- L1 ---- */
uint64_t kmo_kmer_build(int kmer_type, uint64_t val, int k);          /* KmerBuilder::build -> `.0` */
uint64_t kmo_kmer_push(int kmer_type, uint64_t raw, int k, uint8_t base2b);
uint64_t kmo_kmer_revcomp(int kmer_type, uint64_t raw, int k);
uint64_t kmo_kmer_value(int kmer_type, uint64_t raw);                 /* get_compressed_value */
int kmo_kmer_less(int kmer_type, uint64_t a, uint64_t b);             /* Ord */
/* ---- hashes ---- */
uint32_t kmo_int32_hash(uint32_t key);
uint64_t kmo_int64_hash(uint64_t key);
/* minimizer owners / super-k-mer records of a distributed count (the product's own exchange format, kmerutils_amd/csrc/kmu_smer.h) */
uint32_t kmo_minimizer_hash(uint64_t v, int k);
uint32_t kmo_minimizer_owner(uint64_t v, int k, uint32_t n_parts);
void kmo_minimizer_owners(const uint64_t *v, uint64_t n, int k, uint32_t n_parts, uint32_t *out);
uint64_t kmo_superkmer_expand(const uint32_t *recs, uint64_t n_rec, int k, uint64_t *out, int *clean);
double kmo_unif01_f64(uint64_t s[4]); /* one draw of Uniform::<f64>::new(0., 1.) from a xoshiro256++ state */
uint64_t kmo_nohash_finish(uint64_t v, int width_bytes);
uint64_t kmo_fnv1a(uint64_t v, int width_bytes);
uint64_t kmo_nthash_init_8b(const uint8_t *kmer, int k);
uint64_t kmo_nthash_cycle_8b(uint64_t h, int k, uint8_t old_base, uint8_t new_base);
uint64_t kmo_nthash_canonical_init_8b(const uint8_t *kmer, int k, uint64_t *fh, uint64_t *rh, uint8_t *strand);
uint64_t kmo_nthash_canonical_cycle_8b(int k, uint8_t old_base, uint8_t new_base, uint64_t *fh, uint64_t *rh,
                                       uint8_t *strand);
uint64_t kmo_nthash_canonical_2b(uint64_t val, int k, uint64_t *fh, uint64_t *rh, uint8_t *strand);
void kmo_nthash_mult(uint64_t ksize, uint64_t *hashed, int n);
/* ---- RNG (unpinned) ---- */
void kmo_xoshiro_seed(uint64_t seed, uint64_t s[4]);
uint64_t kmo_xoshiro_next(uint64_t s[4]);

/* ---- L2: forward k-mers / fhash, same contract as kmu_kmer_hashes (host memory) ---- */
int kmo_kmer_hashes(const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                    const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *out);

int kmo_kmer_hashes_range(const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                          const uint64_t *packed_offsets, uint32_t n_seq, const uint64_t *range_begin,
                          const uint64_t *range_end, uint64_t *out);
int kmo_kmer_distribution(const kmu_hash_params *p, const uint8_t *bases, const uint64_t *offsets,
                          const uint64_t *packed_offsets, uint32_t n_seq, uint64_t *kmers_out, uint32_t *mult_out,
                          uint64_t cap, uint64_t *dist_offsets_out, uint64_t *n_out);
uint64_t kmo_nthash_rcomp_init_8b(const uint8_t *kmer, int k);
uint64_t kmo_nthash_rcomp_cycle_8b(uint64_t h, int k, uint8_t old_base, uint8_t new_base);
int kmo_nthash(const kmu_nthash_params *p, const uint8_t *bases, const uint64_t *offsets, const uint64_t *packed_offsets,
               uint32_t n_seq, uint64_t *hashes_out, uint8_t *strand_out);

/* ---- L3: same contracts as the kmu_* entry points, host memory only ---- */
int kmo_sketch(const kmu_sketch_params *p, const uint8_t *bases, const uint64_t *offsets,
               const uint64_t *packed_offsets, uint32_t n_seq, const uint64_t *block_row_offsets, void *sig_out,
               uint32_t *counts_out);
int kmo_sketch_hashed(const kmu_sketch_params *p, const void *hashed, const uint64_t *offsets, uint32_t n_seq,
                      void *sig_out, uint32_t *counts_out);
/* ProbMinHash3a on an explicit weighted set (key order = processing order); returns argmin keys and their h */
int kmo_probminhash3a(const uint64_t *keys, const double *weights, uint64_t n, int key_bytes, int m, uint32_t flags,
                      uint64_t *sig_out, double *h_out);

typedef struct kmo_counter kmo_counter;
kmo_counter *kmo_count_create(const kmu_count_params *p);
void kmo_count_destroy(kmo_counter *c);
int kmo_count_add_reads(kmo_counter *c, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq);
int kmo_count_add_kmers(kmo_counter *c, const uint64_t *canon, uint64_t n);
int kmo_count_query(kmo_counter *c, const uint64_t *canon, uint64_t n, uint32_t *counts_out);
uint64_t kmo_count_nb_distinct(kmo_counter *c);
uint64_t kmo_count_nb_unique(kmo_counter *c);
int kmo_count_dump(kmo_counter *c, uint32_t min_count, uint64_t *kmers_out, uint32_t *counts_out, uint64_t cap,
                   uint64_t *n_out);

int kmo_count_once_positions(kmo_counter *c, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, uint64_t *kmers_out,
                             uint32_t *numseq_out, uint32_t *numkmer_out, uint64_t *n_out);

int kmo_probminhash3(const uint64_t *keys, const double *weights, uint64_t n, int key_bytes, int m, uint32_t flags,
                     uint64_t *sig_out);
/* ingest */
int kmo_ingest_fastq(const uint8_t *text, uint64_t n, uint8_t *bases_out, uint64_t *offsets_out, uint32_t *record_index_out,
                     uint64_t info[6]);
/* SetSketch (KMU_ALGO_HLL): parameters of the following kmo_sketch calls (defaults b = 1.001, a = 20, q = 65534); kmo_log is the
 * logarithm both sides use */
void kmo_set_hll_params(double b, double a, uint32_t q);
double kmo_log(double x);
int kmo_ingest_fasta(const uint8_t *text, uint64_t n, uint8_t *bases_out, uint64_t *offsets_out, uint32_t *record_index_out,
                     uint64_t info[6]);
int kmo_ingest_fastx(const uint8_t *text, uint64_t n, uint8_t *bases_out, uint64_t *offsets_out, uint32_t *record_index_out,
                     uint64_t info[6]);
/* signature comparison */
uint32_t kmo_sig_equal_count(const void *a, const void *b, uint32_t m, int word_bytes);
void kmo_minhash_distance(const uint64_t *s1, uint32_t n1, const uint64_t *s2, uint32_t n2, uint32_t out[3]);

#ifdef __cplusplus
}
#endif
#endif
