// kmu_smer.hpp -- host interface of the super-k-mer grouping kernels (kmu_smer.hip) for the distributed counter
// (kmu_count.hip); see kmu_smer.h for the owner function and the record.
#pragma once

#include "kmu_ctx.hpp"
#include "kmu_smer.h"

namespace kmu {

// Duplication sample of a batch: the k-mers whose owner hash (hash owners: bits 8.. of int64_hash(kmer); minimizer owners:
// smer_sampled(minimizer hash)) has `shift` zero bits -- a sample by KEY: every occurrence of a sampled k-mer is in it, so
// occurrences / distinct of the sample estimates the ratio of the batch.  Collected per workgroup in LDS and flushed with
// one global atomic per workgroup.
static constexpr uint32_t SAMPLE_LDS = 4096; // entries per workgroup
struct SampleArgs {
    uint64_t *list;   // null: no sampling
    uint32_t *n;      // [0] entries written, [1] overflow flag
    uint32_t cap;
    uint32_t shift;
};

// what the census of a batch leaves on the device for the scatter (and for the host: the group sizes)
struct SmerGroups {
    int k = 0;
    uint32_t n_parts = 0, units = 0, steps_per_unit = 0;
    void *hist = nullptr;     // u32 [units][n_parts]: records of unit u for owner p
    void *offs = nullptr;     // u64 [units][n_parts]: exclusive prefix over the units
    void *binstart = nullptr; // u64 [n_parts + 1]: first record of every owner's group; [n_parts] = records in all
    void *kmers = nullptr;    // u64 [n_parts]: k-mers in every owner's group
    void *novalid = nullptr;  // bit p set: position p of the flat stream starts no k-mer (k_smer_novalid)
    uint64_t total = 0;       // extent of the flat stream the census walked
};

// "this position starts no k-mer" for every position of a flat stream of reads, 16 bits per aligned code word (bit j of word i:
// position 16 i + j): the last k - 1 positions of every read, what lies before the first read and behind the last up to n_words
// words.  One thread per read; the kernels that walk the stream load a lane's 16 bits next to its 16 bases instead of searching
// the read offsets.  *out: device buffer `buf_name` of the context.
int flat_novalid(kmu_ctx *ctx, const DevSeqs &ds, uint64_t total_bases, int k, uint64_t n_words, const char *buf_name, void **out);

// units of the grouping kernels for a flat stream of total_bases bytes (the sample's capacity is sized from it)
uint32_t smer_units(const kmu_ctx *ctx, uint64_t total_bases);
// census: records and k-mers per owner, validation of the bases, optional duplication sample
int smer_census(kmu_ctx *ctx, const DevSeqs &ds, uint64_t total_bases, int k, uint32_t n_parts, uint32_t *d_err, const SampleArgs &sa,
                SmerGroups *g);
// scatter: the records, grouped by owner in the order of g.binstart, into `records_out` (g.binstart[n_parts] x 12 bytes)
int smer_scatter(kmu_ctx *ctx, const DevSeqs &ds, uint64_t total_bases, const SmerGroups &g, void *records_out);
// k-mers held by n records (sum of their lengths) -> *d_out (device u64; zeroed here)
int smer_count_kmers(kmu_ctx *ctx, const void *records, uint64_t n, uint64_t *d_out);
// records -> canonical k-mers, in no particular order; out holds the n_kmers of smer_count_kmers; d_cursor: device u64 scratch
int smer_expand(kmu_ctx *ctx, const void *records, uint64_t n, int k, uint64_t *out, uint64_t *d_cursor);

} // namespace kmu
