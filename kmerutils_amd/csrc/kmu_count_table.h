// kmu_count_table.h -- the count table of libkmu: where a canonical k-mer lives, what a slot holds, direct insertion and look-up.
// Shared by the three translation units of the counter (kmu_count.hip: the table API; kmu_count_part.hip: the partitioned build;
// kmu_count_dist.hip: the distributed counter).
//
// Reference (src/base/kmercount.rs:241-277): a cuckoo filter holds k-mers seen once, a counting Bloom filter the counts >= 2;
// both are randomised per process, so the observable contract is "exact multiplicity of the canonical k-mer, reported
// saturated at 2^bits - 1" (plus ~3 % false positives the reference itself treats as noise).  Here: one exact table in HBM,
// cut into REGIONS of 4096 slots; a key lives in ONE region, named by its table hash, and is probed inside it along the
// TRIANGULAR sequence home, +1, +3, +6, ... (mod the region size: a permutation of a power of two of slots), so a region is an
// independent little hash table that fits LDS.  (Round 5: linear probing until then.  A region build is paced by the longest
// probe chain among a wave's lanes, and at the load of 0.66 the bench's table now has the clusters of linear probing make that
// 41 trips per region against 25 for the triangular sequence -- 14 / 11 at the old load of 0.44; scripts/sim_region_build.py.
// The first two steps of a chain stay inside the home slot's 128-byte line in HBM.)
//
// Which region (round 5: the number of regions is no longer a power of two).  h = khash(key), a bijection of 64 bits:
//   h = [ g : b1 bits ][ x : 32 bits ][ low : 32 - b1 bits ]
//   group g = the top b1 bits (2^b1 groups: the bins of the first partition level),
//   sub-region s = mulhi32(x, n2) (n2 regions per group, any number up to 2048: the bins of the second level),
//   region = g * n2 + s;  home slot = the top bits of the FRACTION of x * n2 / 2^32, i.e. of the low product word.
// A table of 2^b1 * n2 regions holds what the caller asked for at a load of 0.70 instead of whatever the next power of two
// leaves (0.33 .. 0.67): the bench's table is 48.6 GB instead of 68.7 GB, all of it written by every build.
//
// Two slot formats (kmu_counter::qw):
//  * WIDE, 12 bytes per slot: keys[] + counts[] (small tables).
//  * QUOTIENT, 8 bytes per slot (every big table).  The x values of sub-region s are [lox[s], lox[s + 1]), lox[s] =
//    ceil(s 2^32 / n2): inside its region a key is known by rem = x - lox[s] < 2^(32 - f), f = floor(log2 n2), and `low`.  A slot
//    keeps those 64 - w bits, w = b1 + f, and spends the w bits it saves on the count: slot = (rem : low) << w | count,
//    all-ones = free.  (With n2 a power of two this is "the hash bits below the region index", round 2's form.)  The key of an
//    occupied slot is khash_inv(g : lox[s] + rem : low).  The count field stops at 2^w - 1024 >= 2^counter_bits - 1 (a plain
//    ds_add of the region build may overshoot by the adds in flight, < 512, and never carries into the key bits); every reader
//    clamps to 2^counter_bits - 1 anyway (kmercount.rs:1615).
#pragma once

#include "kmu_comm.hpp"
#include "kmu_ctx.hpp"
#include "kmu_device.h"
#include "kmu_smer.hpp"

struct kmu_counter {
    kmu_ctx *ctx = nullptr;
    kmu_count_params p{};
    uint64_t nslots = 0;  // (2^b1 * n2) << rbits
    int b1 = 0;           // log2 of the number of groups (0: the regions are the n2 sub-regions of one group)
    uint32_t n2 = 1;      // regions per group
    int rbits = 0;        // log2 of the region size
    int qw = 0;           // != 0: quotient format, width of a slot's count field (= b1 + floor(log2 n2)); 0: wide format
    uint64_t *keys = nullptr;    // wide: the keys; quotient: the slots
    uint32_t *counts = nullptr;  // wide only
    uint32_t *lox = nullptr;     // device: [n2 + 1] first x of every sub-region
    uint64_t *scalars = nullptr; // device: [0] distinct, [1] unique, [2] cursor / saturated, [3] occurrences
    bool empty = true;           // table content not materialised yet (every slot is logically free)
    // distributed counters (KMU_COUNT_DISTRIBUTED): one member of a KmerCounterPool spread over the ranks
    bool dist = false;
    int okind = 0;               // owner of a k-mer among the ranks: 0 = the reference's dispatch (intNN_hash(kmer) % n), 1 = its minimizer (kmu_smer.h)
    bool unmerged = false;       // holds entries this rank does not own (MERGE route adds): finalize moves them
    // an exchange in flight between dist_add_begin and dist_add_end
    uint64_t pend_recv = 0;      // k-mers (okind 0) / super-k-mer records (okind 1) arriving in "cnt.recv"
    uint64_t pend_kmers = 0;     // okind 1: the k-mers those records hold
    bool pending = false;
    bool no_seg = false;         // the single-pass partition of this batch has just overflowed: straight to the exact levels
    // KMU_COUNT_HINT_OCCURRENCES: no table yet -- the first add allocates it, sized from the duplication it measures
    bool deferred = false;
};

namespace kmu {

static constexpr uint64_t CKEY_EMPTY = 0xFFFFFFFFFFFFFFFFull; // canonical values are < 2^62
static constexpr int REGION_BITS_MAX = 12;                    // 4096 slots: 32 KiB (quotient) / 48 KiB (wide) of LDS per region
static constexpr uint32_t GROUP_REGIONS_MAX = 2048;           // n2 <= 2048, b1 <= 11: the fan-out of one partition level

__device__ __forceinline__ uint64_t fmix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// Table hash: a bijection of the 64-bit key space.  The complemented form maps the impossible key (all ones) to all
// ones, so CKEY_EMPTY marks "nothing" in the key domain AND in the hash domain.  The partition passes of the build
// carry khash(key) instead of the key: every later digit is a function of the item (no second or third evaluation),
// and the key comes back with khash_inv when it is asked for.
__device__ __forceinline__ uint64_t khash(uint64_t key) { return ~fmix64(~key); }
__device__ __forceinline__ uint64_t khash_inv(uint64_t h) {
    uint64_t x = ~h;
    x ^= x >> 33;
    x *= 0x9cb4b2f8129337dbull; // inverse of 0xc4ceb9fe1a85ec53 mod 2^64
    x ^= x >> 33;
    x *= 0x4f74430c22a54005ull; // inverse of 0xff51afd7ed558ccd mod 2^64
    x ^= x >> 33;
    return ~x;
}

// owner of a k-mer in an n-way key partition: DispatchableT, kmercount.rs:382-420
__device__ __forceinline__ uint64_t owner_hash(uint64_t v, int w32) { return w32 ? (uint64_t) int32_hash((uint32_t) v) : int64_hash(v); }
__device__ __forceinline__ uint32_t owner_of_hash(uint64_t h, int w32, uint32_t n_parts) {
    // 2 / 4 / 8 GPUs: the remainder is a mask (a 64-bit division per k-mer would dominate the grouping kernels)
    if ((n_parts & (n_parts - 1u)) == 0u) return (uint32_t) h & (n_parts - 1u);
    return w32 ? (uint32_t) h % n_parts : (uint32_t) (h % (uint64_t) n_parts);
}
// `mode`: 0 = int64_hash(kmer) % n, 1 = int32_hash(kmer) % n (the reference's dispatch by k-mer width), OWNER_MODE_SMER | k << 8 =
// the minimizer owner of a k-mer of k bases (kmu_smer.h; distributed counters)
static constexpr int OWNER_MODE_SMER = 2;
__host__ __device__ __forceinline__ int owner_mode_smer(int k) { return OWNER_MODE_SMER | (k << 8); }
__device__ __forceinline__ uint32_t kmer_owner(uint64_t v, int mode, uint32_t n_parts) {
    if ((mode & 0xFF) == OWNER_MODE_SMER) return smer_owner_of_kmer(v, mode >> 8, n_parts);
    return owner_of_hash(owner_hash(v, mode), mode, n_parts);
}

// ---- the region map --------------------------------------------------------------------------------------------------------
// One digit of the map as a partition level sees it: n2 == 0: the top bits of the hash (d = h >> sh: the groups, a power-of-two
// level); n2 != 0: d = mulhi32((uint32) (h >> sh), n2) (the sub-regions of a group, any fan-out).
struct Digit {
    int sh;
    uint32_t n2;
};
__device__ __forceinline__ uint32_t digit_of_hash(const Digit &d, uint64_t h) {
    const uint32_t x = (uint32_t) (h >> d.sh);
    return d.n2 ? __umulhi(x, d.n2) : x;
}

struct CountTable {
    uint64_t *keys;      // wide: keys; quotient: slots
    uint32_t *counts;    // wide only
    const uint32_t *lox; // [n2 + 1]
    uint32_t n2;
    int b1;
    int rbits;           // log2 of the region size
    uint32_t rmask;      // region size - 1
    int w;               // quotient format: width of the count field; 0 = wide format
};

// where a hash lives: its region's first slot, its home slot inside the region, and (x, s) for the quotient
struct TabLoc {
    uint64_t base;
    uint32_t off, x, s;
};
__device__ __forceinline__ TabLoc tab_locate(const CountTable &t, uint64_t h) {
    TabLoc l;
    l.x = (uint32_t) (h >> (32 - t.b1));
    l.s = __umulhi(l.x, t.n2);
    const uint32_t g = t.b1 ? (uint32_t) (h >> (64 - t.b1)) : 0u;
    l.base = (uint64_t) (g * t.n2 + l.s) << t.rbits;
    l.off = (l.x * t.n2) >> (32 - t.rbits);
    return l;
}
__device__ __forceinline__ uint32_t region_of_hash(const CountTable &t, uint64_t h) { return (uint32_t) (tab_locate(t, h).base >> t.rbits); }

// quotient slots: (rem : low) << w | count
static constexpr uint32_t Q_MARGIN = 1024; // a count field stops at 2^w - Q_MARGIN (+ the plain adds in flight of a region build, < 512)
__device__ __forceinline__ uint64_t q_cmask(int w) { return (1ull << w) - 1ull; }
__device__ __forceinline__ uint64_t q_limit(int w) { return (1ull << w) - (uint64_t) Q_MARGIN; }
__device__ __forceinline__ bool q_same(uint64_t slot, uint64_t hw, int w) { return ((slot ^ hw) >> w) == 0ull; }
// the 64 - w bits a slot keeps of a hash whose sub-region starts at x = lox_s
__device__ __forceinline__ uint64_t q_kept(int b1, uint64_t h, uint32_t x, uint32_t lox_s) {
    return ((uint64_t) (x - lox_s) << (32 - b1)) | (h & ((1ull << (32 - b1)) - 1ull));
}
__device__ __forceinline__ uint64_t q_hw(const CountTable &t, uint64_t h, const TabLoc &l) { return q_kept(t.b1, h, l.x, t.lox[l.s]) << t.w; }
__device__ __forceinline__ uint64_t q_key_of(const CountTable &t, uint64_t idx, uint64_t slot) {
    const uint32_t region = (uint32_t) (idx >> t.rbits), g = region / t.n2, s = region - g * t.n2;
    const uint64_t h = (t.b1 ? (uint64_t) g << (64 - t.b1) : 0ull) + (((uint64_t) t.lox[s] << (32 - t.b1)) + (slot >> t.w));
    return khash_inv(h);
}

// slot i of either format: false = free; `cnt` 0 = an entry that left for its owner (distributed counters)
template <bool WANT_KEY>
__device__ __forceinline__ bool slot_read(const CountTable &t, uint64_t i, uint64_t &key, uint32_t &cnt) {
    const uint64_t s = t.keys[i];
    if (s == CKEY_EMPTY) return false;
    if (t.w) {
        const uint64_t f = s & q_cmask(t.w), lim = q_limit(t.w);
        cnt = (uint32_t) (f < lim ? f : lim); // (a region build may leave a field a few hundred above its ceiling)
        if (WANT_KEY) key = q_key_of(t, i, s);
    } else {
        cnt = t.counts[i];
        key = s;
    }
    return true;
}
__device__ __forceinline__ void slot_zero_count(const CountTable &t, uint64_t i) { // (one thread per slot, nothing concurrent)
    if (t.w) t.keys[i] &= ~q_cmask(t.w);
    else t.counts[i] = 0u;
}

// KmerCounter::insert_kmer (kmercount.rs:241-267), exact form: count[v] += add.  h = khash(v); the wide format compares
// keys (v), the quotient format the hash bits a slot keeps (v is not looked at).
__device__ __forceinline__ bool count_insert_h(const CountTable &t, uint64_t v, uint64_t h, uint32_t add) {
    const TabLoc l = tab_locate(t, h);
    uint32_t off = l.off;
    if (t.w) {
        const uint64_t hw = q_hw(t, h, l), cmask = q_cmask(t.w), limit = q_limit(t.w);
        const uint64_t a = add < limit ? add : limit;
        for (uint32_t probes = 0; probes <= t.rmask; probes++) {
            const uint64_t idx = l.base | off;
            uint64_t cur = __hip_atomic_load(&t.keys[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == CKEY_EMPTY) {
                cur = atomicCAS((unsigned long long *) &t.keys[idx], (unsigned long long) CKEY_EMPTY, (unsigned long long) (hw | a));
                if (cur == CKEY_EMPTY) return true;
            }
            if (q_same(cur, hw, t.w)) { // saturating add (the key bits of an occupied slot never change)
                for (;;) {
                    const uint64_t cnt = cur & cmask;
                    if (cnt >= limit) return true;
                    const uint64_t n = cnt + a < limit ? cnt + a : limit;
                    const uint64_t prev = atomicCAS((unsigned long long *) &t.keys[idx], (unsigned long long) cur, (unsigned long long) ((cur & ~cmask) | n));
                    if (prev == cur) return true;
                    cur = prev;
                }
            }
            off = (off + probes + 1u) & t.rmask;
        }
        return false;
    }
    for (uint32_t probes = 0; probes <= t.rmask; probes++) {
        const uint64_t idx = l.base | off;
        uint64_t cur = __hip_atomic_load(&t.keys[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == CKEY_EMPTY) {
            cur = atomicCAS((unsigned long long *) &t.keys[idx], (unsigned long long) CKEY_EMPTY, (unsigned long long) v);
            if (cur == CKEY_EMPTY) cur = v;
        }
        if (cur == v) {
            atomicAdd(&t.counts[idx], add);
            return true;
        }
        off = (off + probes + 1u) & t.rmask;
    }
    return false;
}
__device__ __forceinline__ bool count_insert(const CountTable &t, uint64_t v, uint32_t add) { return count_insert_h(t, v, khash(v), add); }

__device__ __forceinline__ uint32_t count_lookup(const CountTable &t, uint64_t v) {
    const uint64_t h = khash(v);
    const TabLoc l = tab_locate(t, h);
    uint32_t off = l.off;
    const uint64_t hw = t.w ? q_hw(t, h, l) : 0ull;
    for (uint32_t probes = 0; probes <= t.rmask; probes++) {
        const uint64_t idx = l.base | off;
        uint64_t cur = t.keys[idx];
        if (cur == CKEY_EMPTY) return 0;
        if (t.w) {
            if (q_same(cur, hw, t.w)) {
                const uint64_t f = cur & q_cmask(t.w), lim = q_limit(t.w);
                return (uint32_t) (f < lim ? f : lim);
            }
        } else if (cur == v) return t.counts[idx];
        off = (off + probes + 1u) & t.rmask;
    }
    return 0;
}

// ---- host side shared by the translation units -------------------------------------------------------------------------------
inline uint64_t table_regions(const kmu_counter *c) { return c->nslots >> c->rbits; }
inline size_t table_image_bytes(const kmu_counter *c) { return (size_t) c->nslots * (c->qw ? 8 : 12); }
inline CountTable table_of(const kmu_counter *c) {
    CountTable t;
    t.keys = c->keys;
    t.counts = c->counts;
    t.lox = c->lox;
    t.n2 = c->n2;
    t.b1 = c->b1;
    t.rbits = c->rbits;
    t.rmask = (1u << c->rbits) - 1u;
    t.w = c->qw;
    return t;
}
inline uint32_t max_count(const kmu_counter *c) { return c->p.counter_bits == 8 ? 255u : 65535u; }
inline int grid_for(const kmu_ctx *ctx, uint64_t n, int per_block) {
    const uint64_t blocks = (n + per_block - 1) / per_block, cap = (uint64_t) ctx->num_cus * 8;
    return (int) (blocks < 1 ? 1 : blocks < cap ? blocks : cap);
}
// kmer_owner's mode for the counter's own owner function (distributed counters)
inline int owner_mode_of(const kmu_counter *c) { return c->okind == 1 ? owner_mode_smer(c->p.kmer_size) : (kmer_val_bytes(c->p.kmer_type) == 4 ? 1 : 0); }

// the partition plan of a table: which digits its levels sort by (kmu_count_part.hip)
struct PartPlan {
    int b1;          // log2 of the level-1 fan-out of a two-level plan (the table's groups); 0: ONE level, by sub-region
    uint32_t n2;     // fan-out of the last level (the table's sub-regions per group)
    uint32_t units1; // level-1 units (workgroups), each a contiguous range of wave steps
    uint32_t steps_per_unit;
    uint32_t chunks2;     // level-2 units per level-1 partition (exact route)
    uint32_t owner_parts; // != 0: level 1 groups the k-mers by owner rank instead (multi-GPU exchange), bins1 = owner_parts
    int owner_w32;
};
__host__ __device__ __forceinline__ uint32_t plan_bins1(const PartPlan &pl) { return pl.owner_parts ? pl.owner_parts : pl.b1 ? 1u << pl.b1 : pl.n2; }
__host__ __device__ __forceinline__ Digit plan_digit1(const PartPlan &pl) { return pl.b1 ? Digit{64 - pl.b1, 0u} : Digit{32, pl.n2}; }
__host__ __device__ __forceinline__ Digit plan_digit2(const PartPlan &pl) { return Digit{32 - pl.b1, pl.n2}; }
inline uint64_t plan_regions(const PartPlan &pl) { return ((uint64_t) 1 << pl.b1) * pl.n2; }

// ---- cross-unit entry points ---------------------------------------------------------------------------------------------------
// kmu_count.hip
int table_alloc(kmu_counter *c, uint64_t distinct_hint);
int table_alloc_for(kmu_counter *c, const DevSeqs *ds, uint64_t total_bases, uint32_t *d_err, double known_ratio = 0.0, uint64_t occurrences = 0);
int materialize(kmu_counter *c);
int local_add(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t *d_err);
int add_entries(kmu_counter *c, const uint64_t *kmers, const uint32_t *counts, uint64_t n, int mem);
int add_superkmers(kmu_counter *c, const void *recs, uint64_t n_rec, uint64_t n_kmers);
int select_entries(kmu_counter *c, uint32_t min_count, uint32_t maxc, uint32_t part, uint32_t n_parts, uint64_t *kmers_out, uint32_t *counts_out,
                   uint64_t cap, int mem, bool sort, uint64_t *n_out, bool own_owner = false);
int flat_stream_extent(kmu_ctx *ctx, const uint64_t *host_offsets, uint32_t n_seq, int mem, DevSeqs &ds, uint64_t *total_out);
bool partitioned_batch_wanted(const kmu_counter *c, uint64_t n);
// kmu_count_part.hip
bool part_plan_for(const kmu_counter *c, PartPlan *pl); // false: the table has more regions than two levels reach
bool seg_partition_wanted(uint64_t n_items);
int partitioned_add(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t *d_err);
int partitioned_add_kmers(kmu_counter *c, const uint64_t *d_kmers, uint64_t n, uint32_t *d_err);
int seg_partitioned_add_kmers(kmu_counter *c, const uint64_t *d_kmers, uint64_t n, const PartPlan &pl, uint32_t *d_err, int *taken,
                              const void *recs = nullptr, uint64_t n_rec = 0);
// the canonical k-mers of the reads grouped by owner rank: census (per-unit histogram of the owners + the duplication sample), scatter
struct OwnerPlan {
    PartPlan pl;
    void *hist1, *offs1, *tot1, *binstart1;
};
int owner_census(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t n_parts, uint32_t *d_err, OwnerPlan *op, const SampleArgs &sa);
int owner_scatter(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, const OwnerPlan &op, uint64_t **dev_out);
int sample_ratio(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t *d_err, double *ratio_out);
int sample_distinct(kmu_ctx *ctx, const uint64_t *d_list, uint32_t n, uint32_t *d_scratch_word, uint32_t *distinct_out);
// kmu_count_dist.hip
int dist_add_begin(kmu_counter *c, DevSeqs &ds, uint64_t total_bases, uint32_t *d_err);
int dist_add_end(kmu_counter *c);

} // namespace kmu
