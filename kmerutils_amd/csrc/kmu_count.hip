// kmu_count.hip -- k-mer counting on gfx950 behind the KmerCountT contract.
//
// Reference (src/base/kmercount.rs:241-277): a cuckoo filter holds k-mers seen once, a counting Bloom filter
// the counts >= 2; both are randomised per process, so the observable contract is "exact multiplicity of the
// canonical k-mer, reported saturated at 2^bits - 1" (plus ~3 % false positives the reference itself treats as
// noise).  Here: one exact table in HBM -- keys[2^q] (canonical value, all-ones = free) + counts[2^q] (u32) --
// cut into REGIONS of 4096 slots; a key lives in the region named by the top bits of khash(key) and is probed
// linearly inside it (wrapping at the region end), so a region is an independent little hash table that fits LDS.
//
// Two ways to add k-mers:
//  * big batches (the throughput path): a radix-partitioned build.  The bases are consumed as ONE flat stream of
//    aligned 16-byte words; exact per-unit histograms give every workgroup private, contiguous output ranges, so
//    the k-mers are scattered region by region (one or two levels, <= 2048-way each) with plain stores and no
//    global atomics; then one workgroup per region builds the region in LDS (ds_cmpst / ds_add) and streams it
//    out.  HBM sees only streaming reads/writes: bases, 8 B per k-mer per level, and the table image.
//  * small batches / single values / merges: direct insertion with one 64-bit CAS + one 32-bit atomic add.
//
// Two slot formats (kmu_counter::qw):
//  * WIDE, 12 bytes per slot: keys[] + counts[] as above (small tables).
//  * QUOTIENT, 8 bytes per slot (tables of >= 2^23 / 2^29 slots for 8- / 16-bit counters: every big table): khash is a
//    bijection and the region index IS its top w = lg - 12 bits, so a slot keeps only the other 64 - w bits and spends the w
//    bits it saves on the count: slot = (khash(key) << w) | count, all-ones = free.  The key of an occupied slot is
//    khash_inv(region << (64 - w) | slot >> w).  The count field stops at LIMIT = 2^w - 1024 >= 2^counter_bits - 1 (a plain
//    ds_add of the region build may overshoot by the adds in flight, < 512, and never carries into the key bits); every
//    reader clamps to 2^counter_bits - 1 anyway (kmercount.rs:1615).  The region of a build is 32 KiB of LDS instead of 48
//    (four workgroups per CU), a first sighting is ONE ds_cmpst_rtn_b64, the LDS image is the HBM image: 68.7 GB instead of
//    103 GB written per build of the bench workload.
#include <algorithm>
#include <vector>

#include "kmu_comm.hpp"
#include "kmu_ctx.hpp"
#include "kmu_flat.h"
#include "kmu_smer.hpp"
#include "kmu_stream.h"

// How level 1 of the single-pass partition learns where reads end: 2 = from the "no k-mer" bits of flat_novalid (16 bits per lane,
// requested with its 16 bases; round 4), 1 = from a table of the wave steps' reads (k_step_table, round 3), 0 = by looking the read
// of a step up at the top of the step (A/B builds)
#ifndef KMU_STEP_TAB
#define KMU_STEP_TAB 2
#endif
#ifndef KMU_SCATTER_NT
#define KMU_SCATTER_NT 0
#endif
#ifndef KMU_SCATTER_NTLOAD // A/B builds: 1 = the level-2 input, 2 = the level-1 bases as well, with non-temporal loads
#define KMU_SCATTER_NTLOAD 0
#endif

struct kmu_counter {
    kmu_ctx *ctx = nullptr;
    kmu_count_params p{};
    uint64_t nslots = 0; // power of two
    int lg = 0;
    int rbits = 0;       // log2 of the region size
    int qw = 0;          // != 0: quotient format, width of a slot's count field (= lg - rbits); 0: wide format
    uint64_t *keys = nullptr;    // wide: the keys; quotient: the slots
    uint32_t *counts = nullptr;  // wide only
    uint64_t *scalars = nullptr; // device: [0] distinct, [1] unique, [2] cursor
    bool empty = true;           // table content not materialised yet (every slot is logically free)
    // COMPACT state (what a partitioned build leaves): region r holds its rcount[r] (key, count) pairs at the head of its
    // slab, nothing else of the slab is written or meaningful.  Readers that probe slots first expand it (materialize).
    bool compact = false;
    uint32_t *rcount = nullptr;  // entries per region (compact state)
    bool stats_cached = false;   // scalars[4..6] = distinct / unique / occurrences, left by the last compact build
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
static constexpr int REGION_BITS_MAX = 12;                    // 4096 slots = 48 KiB of LDS per region

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
// carry khash(key) instead of the key: every later digit is a bit field of the item (no second or third evaluation),
// and the key comes back with khash_inv when the region is built.
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

struct CountTable {
    uint64_t *keys;   // wide: keys; quotient: slots
    uint32_t *counts; // wide only
    int shift;        // 64 - lg
    uint32_t rmask;   // region size - 1
    int w;            // quotient format: width of the count field (the region bits of the table); 0 = wide format
    int rbits;        // log2 of the region size
};

// quotient slots: (khash(key) << w) | count
static constexpr uint32_t Q_MARGIN = 1024; // a count field stops at 2^w - Q_MARGIN (+ the plain adds in flight of a region build, < 512)
__device__ __forceinline__ uint64_t q_cmask(int w) { return (1ull << w) - 1ull; }
__device__ __forceinline__ uint64_t q_limit(int w) { return (1ull << w) - (uint64_t) Q_MARGIN; }
__device__ __forceinline__ bool q_same(uint64_t slot, uint64_t hw, int w) { return ((slot ^ hw) >> w) == 0ull; }
__device__ __forceinline__ uint64_t q_key_of(const CountTable &t, uint64_t idx, uint64_t slot) {
    return khash_inv(((idx >> t.rbits) << (64 - t.w)) | (slot >> t.w));
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
    const uint64_t idx0 = h >> t.shift;
    const uint64_t base = idx0 & ~(uint64_t) t.rmask;
    uint32_t off = (uint32_t) idx0 & t.rmask;
    if (t.w) {
        const uint64_t hw = h << t.w, cmask = q_cmask(t.w), limit = q_limit(t.w);
        const uint64_t a = add < limit ? add : limit;
        for (uint32_t probes = 0; probes <= t.rmask; probes++) {
            const uint64_t idx = base | off;
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
            off = (off + 1) & t.rmask;
        }
        return false;
    }
    for (uint32_t probes = 0; probes <= t.rmask; probes++) {
        const uint64_t idx = base | off;
        uint64_t cur = __hip_atomic_load(&t.keys[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == CKEY_EMPTY) {
            cur = atomicCAS((unsigned long long *) &t.keys[idx], (unsigned long long) CKEY_EMPTY, (unsigned long long) v);
            if (cur == CKEY_EMPTY) cur = v;
        }
        if (cur == v) {
            atomicAdd(&t.counts[idx], add);
            return true;
        }
        off = (off + 1) & t.rmask;
    }
    return false;
}
__device__ __forceinline__ bool count_insert(const CountTable &t, uint64_t v, uint32_t add) { return count_insert_h(t, v, khash(v), add); }

__device__ __forceinline__ uint32_t count_lookup(const CountTable &t, uint64_t v) {
    const uint64_t h = khash(v);
    const uint64_t idx0 = h >> t.shift;
    const uint64_t base = idx0 & ~(uint64_t) t.rmask;
    uint32_t off = (uint32_t) idx0 & t.rmask;
    const uint64_t hw = t.w ? h << t.w : 0ull;
    for (uint32_t probes = 0; probes <= t.rmask; probes++) {
        const uint64_t idx = base | off;
        uint64_t cur = t.keys[idx];
        if (cur == CKEY_EMPTY) return 0;
        if (t.w) {
            if (q_same(cur, hw, t.w)) {
                const uint64_t f = cur & q_cmask(t.w), lim = q_limit(t.w);
                return (uint32_t) (f < lim ? f : lim);
            }
        } else if (cur == v) return t.counts[idx];
        off = (off + 1) & t.rmask;
    }
    return 0;
}

// One wave step (64 words = 1024 bases) of the flat base stream: f(canon) for every k-mer that lies inside one read.
// Returns a non-zero mask if this lane saw a non-ACGT byte.
// The reads occupy [offsets[0], total) of the stream: `start` = offsets[0] need not be 0 (a range of a larger read set).
template <typename F>
__device__ __forceinline__ uint32_t flat_step_canon(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq,
                                                    uint64_t total, uint64_t start, int k, uint64_t st, uint32_t &r_hint, F &&f) {
    SeqView s;
    s.base = bases; s.begin = 0; s.len = total; s.total = total; s.packed = 0;
    const int lane = lane_id();
    const uint64_t widx = st * 64 + lane;
    uint32_t bad, bad2;
    uint32_t w0 = load_code_word(s, widx, bad);
    uint32_t ex = load_code_word(s, st * 64 + 64 + (uint64_t) (lane & 1), bad2);
    uint32_t e0 = bcast_u32(ex, 0), e1 = bcast_u32(ex, 1);
    uint32_t w1 = shfl_down_u32(w0, 1), w2 = shfl_down_u32(w0, 2);
    if (lane == 63) { w1 = e0; w2 = e1; }
    if (lane == 62) { w2 = e0; }
    uint32_t r = wave_find_read_from(offsets, n_seq, st * 1024 < total ? st * 1024 : total - 1, r_hint);
    r_hint = r;
    const uint64_t g0 = widx * 16;
    const bool in = g0 < total && g0 + 16 > start;
    uint64_t rend = 0;
    if (in) {
        rend = offsets[r + 1];
        while (g0 >= rend && r + 1 < n_seq) { r++; rend = offsets[r + 1]; } // the read of this lane's first base
    }
    const uint64_t hi = ((uint64_t) w0 << 32) | w1;
    const int sh = 64 - 2 * k;
    // A wave whose lanes all sit well inside a read (long reads: most waves) skips the per-k-mer boundary tests.
    if (__all(!in || (g0 >= start && rend - g0 >= (uint64_t) (15 + k)))) {
        if (in) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                uint64_t v = (hi << (2 * j)) | (((uint64_t) w2 << (2 * j)) >> 32);
                uint64_t val = v >> sh;
                uint64_t rc = revcomp_val(val, k);
                f(rc < val ? rc : val); // kmer.reverse_complement().min(kmer), kmercount.rs:938
            }
        }
    } else if (in) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint64_t g = g0 + j;
            while (g >= rend && r + 1 < n_seq) { r++; rend = offsets[r + 1]; }
            if (g >= start && g + k <= rend) {
                uint64_t v = (hi << (2 * j)) | (((uint64_t) w2 << (2 * j)) >> 32);
                uint64_t val = v >> sh;
                uint64_t rc = revcomp_val(val, k);
                f(rc < val ? rc : val);
            }
        }
    }
    return bad;
}

// ------------------------------------------------------------------------------------------------
// direct insertion kernels (small batches, packed input, explicit k-mer lists, merges)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_count_add_flat(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq,
                                                        int k, CountTable t, uint32_t *err) {
    const uint64_t total = offsets[n_seq], start = offsets[0];
    const uint64_t nsteps = ((total + 15) / 16 + 63) / 64;
    const uint64_t wave_global = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves_global = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    uint32_t anybad = 0, full = 0, r_hint = 0xFFFFFFFFu;
    for (uint64_t st = wave_global; st < nsteps; st += nwaves_global)
        anybad |= flat_step_canon(bases, offsets, n_seq, total, start, k, st, r_hint, [&](uint64_t canon) {
            if (!count_insert(t, canon, 1u)) full = 1;
        });
    if (anybad) atomicOr(err, DERR_NON_ACGT);
    if (full) atomicOr(err, DERR_TABLE_FULL);
}

// ---- k-mers seen exactly once, with their positions (KmerFilter1, src/base/kmercount.rs:985-1082) ----------------------
// One wave step of the flat stream: bit j of the returned mask = "the k-mer starting at this lane's base j lies inside one
// read and its canonical value has count 1"; canon[j], and the read of base j in rd[j], are filled for those bits.
__device__ __forceinline__ uint32_t once_step(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, uint64_t total,
                                              uint64_t start, int k, uint64_t st, uint32_t &r_hint, const CountTable &t,
                                              uint64_t (&canon)[16], uint32_t (&rd)[16]) {
    SeqView s;
    s.base = bases; s.begin = 0; s.len = total; s.total = total; s.packed = 0;
    const int lane = lane_id();
    const uint64_t widx = st * 64 + lane;
    uint32_t bad, bad2;
    uint32_t w0 = load_code_word(s, widx, bad);
    uint32_t ex = load_code_word(s, st * 64 + 64 + (uint64_t) (lane & 1), bad2);
    uint32_t e0 = bcast_u32(ex, 0), e1 = bcast_u32(ex, 1);
    uint32_t w1 = shfl_down_u32(w0, 1), w2 = shfl_down_u32(w0, 2);
    if (lane == 63) { w1 = e0; w2 = e1; }
    if (lane == 62) { w2 = e0; }
    uint32_t r = wave_find_read_from(offsets, n_seq, st * 1024 < total ? st * 1024 : total - 1, r_hint);
    r_hint = r;
    const uint64_t g0 = widx * 16;
    uint32_t mask = 0;
    if (g0 < total && g0 + 16 > start) {
        uint64_t rend = offsets[r + 1];
        const uint64_t hi = ((uint64_t) w0 << 32) | w1;
        const int sh = 64 - 2 * k;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint64_t g = g0 + j;
            while (g >= rend && r + 1 < n_seq) { r++; rend = offsets[r + 1]; }
            if (g >= start && g + k <= rend) {
                const uint64_t v = (hi << (2 * j)) | (((uint64_t) w2 << (2 * j)) >> 32);
                const uint64_t val = v >> sh, rc = revcomp_val(val, k);
                const uint64_t c = rc < val ? rc : val;
                if (count_lookup(t, c) == 1u) {
                    mask |= 1u << j;
                    canon[j] = c;
                    rd[j] = r;
                }
            }
        }
    }
    return mask;
}

__global__ void __launch_bounds__(256) k_once_count(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int k,
                                                    CountTable t, uint32_t *cnt) {
    const uint64_t total = offsets[n_seq], start = offsets[0];
    const uint64_t nsteps = ((total + 15) / 16 + 63) / 64;
    const uint64_t wave_global = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves_global = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    uint32_t r_hint = 0xFFFFFFFFu;
    for (uint64_t st = wave_global; st < nsteps; st += nwaves_global) {
        uint64_t canon[16];
        uint32_t rd[16];
        const uint32_t mask = once_step(bases, offsets, n_seq, total, start, k, st, r_hint, t, canon, rd);
        const uint32_t incl = wave_incl_scan_u32((uint32_t) __popc(mask));
        if (lane_id() == 63) cnt[st] = incl;
    }
}

// the same walk; step st writes its records at base[st] .. in (sequence, position) order
__global__ void __launch_bounds__(256) k_once_emit(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int k,
                                                   CountTable t, const uint64_t *base, uint64_t *kmers_out,
                                                   uint32_t *numseq_out, uint32_t *numkmer_out) {
    const uint64_t total = offsets[n_seq], start = offsets[0];
    const uint64_t nsteps = ((total + 15) / 16 + 63) / 64;
    const uint64_t wave_global = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves_global = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    uint32_t r_hint = 0xFFFFFFFFu;
    for (uint64_t st = wave_global; st < nsteps; st += nwaves_global) {
        uint64_t canon[16];
        uint32_t rd[16];
        const uint32_t mask = once_step(bases, offsets, n_seq, total, start, k, st, r_hint, t, canon, rd);
        const uint32_t c = (uint32_t) __popc(mask);
        uint64_t at = base[st] + (wave_incl_scan_u32(c) - c);
        const uint64_t g0 = (st * 64 + (uint64_t) lane_id()) * 16;
#pragma unroll
        for (int j = 0; j < 16; j++)
            if ((mask >> j) & 1u) {
                kmers_out[at] = canon[j];
                numseq_out[at] = rd[j];
                numkmer_out[at] = (uint32_t) (g0 + j - offsets[rd[j]]);
                at++;
            }
    }
}

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

// [0] += occupied slots, [1] += slots with count == 1, [3] += sum of the counts (k-mer occurrences held)
// scalars[0] distinct, [1] unique, [2] slots whose count sits at (or above) `ceiling`, [3] occurrences
__global__ void __launch_bounds__(256) k_count_stats(CountTable t, uint64_t nslots, uint64_t *scalars, uint32_t ceiling) {
    uint64_t d = 0, u = 0, tot = 0, sat = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (uint64_t) gridDim.x * blockDim.x) {
        uint64_t key;
        uint32_t c; // 0: an entry that left for its owner (distributed counters)
        if (slot_read<false>(t, i, key, c)) {
            d += c != 0u;
            u += c == 1u;
            tot += c;
            sat += c >= ceiling;
        }
    }
    if (__any(sat != 0)) {
        for (int o = 32; o >= 1; o >>= 1) sat += (uint64_t) (uint32_t) __shfl_xor((int) (uint32_t) sat, o, 64);
        if (lane_id() == 0) atomicAdd((unsigned long long *) &scalars[2], (unsigned long long) sat);
    }
    for (int o = 32; o >= 1; o >>= 1) {
        d += ((uint64_t) (uint32_t) __shfl_xor((int) (d >> 32), o, 64) << 32) | (uint32_t) __shfl_xor((int) (uint32_t) d, o, 64);
        u += ((uint64_t) (uint32_t) __shfl_xor((int) (u >> 32), o, 64) << 32) | (uint32_t) __shfl_xor((int) (uint32_t) u, o, 64);
        tot += ((uint64_t) (uint32_t) __shfl_xor((int) (tot >> 32), o, 64) << 32) | (uint32_t) __shfl_xor((int) (uint32_t) tot, o, 64);
    }
    if (lane_id() == 0) {
        if (d) atomicAdd((unsigned long long *) &scalars[0], (unsigned long long) d);
        if (u) atomicAdd((unsigned long long *) &scalars[1], (unsigned long long) u);
        if (tot) atomicAdd((unsigned long long *) &scalars[3], (unsigned long long) tot);
    }
}

// compact (kmer, count) with count >= min_count (and owner == part when n_parts > 0); cap-limited
__global__ void __launch_bounds__(256) k_count_select(CountTable t, uint64_t nslots, uint32_t min_count, uint32_t maxc,
                                                      int w32, uint32_t part, uint32_t n_parts, uint64_t cap,
                                                      uint64_t *kmers, uint32_t *counts, uint64_t *cursor) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    const uint64_t rounds = (nslots + stride - 1) / stride; // wave-uniform trip count (ballot inside)
    for (uint64_t it = 0; it < rounds; it++) {
        const uint64_t i = it * stride + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
        bool take = false;
        uint64_t key = 0;
        uint32_t c = 0;
        if (i < nslots && slot_read<true>(t, i, key, c)) {
            // (a zero count: an entry that left for its owner at a MERGE finalize -- absent for every reader)
            take = c != 0u && c >= min_count && !(n_parts && kmer_owner(key, w32, n_parts) != part);
        }
        // one global atomic per wave
        const uint64_t m = __ballot(take);
        if (m) {
            const int leader = __ffsll((unsigned long long) m) - 1;
            uint64_t base = 0;
            if (lane_id() == leader) base = atomicAdd((unsigned long long *) cursor, (unsigned long long) __popcll(m));
            base = ((uint64_t) bcast_u32((uint32_t) (base >> 32), leader) << 32) | bcast_u32((uint32_t) base, leader);
            if (take) {
                uint64_t pos = base + (uint64_t) __popcll(m & ((1ull << lane_id()) - 1ull));
                if (kmers && pos < cap) {
                    kmers[pos] = key;
                    counts[pos] = c > maxc ? maxc : c;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// radix-partitioned build
// ------------------------------------------------------------------------------------------------
struct PartPlan {
    int region_bits; // number of regions = 2^region_bits
    int b1, b2;      // level-1 / level-2 fan-out bits (b2 = 0: one level)
    uint32_t units1; // level-1 units (workgroups), each a contiguous range of wave steps
    uint32_t steps_per_unit;
    uint32_t chunks2; // level-2 units per level-1 partition
    uint32_t owner_parts; // != 0: level 1 groups the k-mers by owner rank instead (multi-GPU exchange), bins1 = owner_parts
    int owner_w32;
};

__device__ __forceinline__ uint32_t region_of_hash(uint64_t h, int region_bits) {
    return region_bits ? (uint32_t) (h >> (64 - region_bits)) : 0u;
}
__device__ __forceinline__ uint32_t region_of(uint64_t canon, int region_bits) {
    return region_of_hash(khash(canon), region_bits);
}

// level 1, pass 1: per-unit histogram of the level-1 digit (also validates the bases)
__global__ void __launch_bounds__(256) k_part_hist1(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int k,
                                                    PartPlan pl, uint32_t *hist1, uint32_t *err, SampleArgs sa) {
    extern __shared__ uint32_t lh[];
    const uint32_t bins1 = pl.owner_parts ? pl.owner_parts : 1u << pl.b1;
    // sampling (owner grouping only): list and counter behind the histogram, 8-byte aligned
    uint32_t *ls_n = lh + ((bins1 + 1u) & ~1u);
    uint64_t *ls = reinterpret_cast<uint64_t *>(ls_n + 2);
    for (uint32_t b = threadIdx.x; b < bins1; b += blockDim.x) lh[b] = 0;
    if (sa.list && threadIdx.x == 0) ls_n[0] = 0;
    __syncthreads();
    const uint64_t total = offsets[n_seq], start = offsets[0];
    const uint64_t nsteps = ((total + 15) / 16 + 63) / 64;
    const uint64_t s0 = (uint64_t) blockIdx.x * pl.steps_per_unit;
    const uint64_t s1 = s0 + pl.steps_per_unit < nsteps ? s0 + pl.steps_per_unit : nsteps;
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint64_t smask = sa.shift >= 32 ? 0xFFFFFFFFull : ((1ull << sa.shift) - 1ull);
    uint32_t bad = 0, r_hint = 0xFFFFFFFFu;
    // up to eight owners (one node's GPUs): a lane counts its 16 k-mers of a wave step in eight 8-bit fields of a register and
    // the wave adds its sums to the histogram with eight atomics per step (round 2 took one LDS atomic per k-mer on eight
    // addresses: 64 lanes on 8 words, 15 ms for the bench shard)
    const bool packed = pl.owner_parts != 0 && pl.owner_parts <= 8;
    for (uint64_t st = s0 + wave; st < s1; st += nwaves) {
        uint64_t pc = 0;
        bad |= flat_step_canon(bases, offsets, n_seq, total, start, k, st, r_hint, [&](uint64_t canon) {
            if (pl.owner_parts) {
                const uint64_t h = owner_hash(canon, pl.owner_w32);
                const uint32_t o = owner_of_hash(h, pl.owner_w32, pl.owner_parts);
                if (packed) pc += 1ull << (8u * o);
                else atomicAdd(&lh[o], 1u);
                if (sa.list && ((h >> 8) & smask) == 0ull) {
                    const uint32_t at = atomicAdd(&ls_n[0], 1u);
                    if (at < SAMPLE_LDS) ls[at] = canon;
                }
            } else {
                atomicAdd(&lh[region_of(canon, pl.region_bits) >> pl.b2], 1u);
            }
        });
        if (packed) { // (wave-uniform) fields 0 2 4 6 and 1 3 5 7 as 16-bit numbers, two to a word: a wave's sums stay below 2^16
            const uint64_t ev = pc & 0x00FF00FF00FF00FFull, od = (pc >> 8) & 0x00FF00FF00FF00FFull;
            const uint32_t w4[4] = {(uint32_t) ev, (uint32_t) (ev >> 32), (uint32_t) od, (uint32_t) (od >> 32)};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t sum = wave_incl_scan_u32(w4[i]); // (no carry between the halves: each stays below 2^16)
                if (lane_id() == 63) {
                    const uint32_t f0 = (uint32_t) (i & 1) * 4u + (uint32_t) (i >> 1); // owner of the low half: 0, 4, 1, 5
                    const uint32_t lo = sum & 0xFFFFu, hi = sum >> 16;
                    if (lo && f0 < bins1) atomicAdd(&lh[f0], lo);
                    if (hi && f0 + 2u < bins1) atomicAdd(&lh[f0 + 2u], hi);
                }
            }
        }
    }
    if (bad) atomicOr(err, DERR_NON_ACGT);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < bins1; b += blockDim.x) hist1[(uint64_t) blockIdx.x * bins1 + b] = lh[b];
    if (sa.list) {
        __shared__ uint32_t gbase;
        const uint32_t cnt = ls_n[0], keep = cnt < SAMPLE_LDS ? cnt : SAMPLE_LDS;
        if (threadIdx.x == 0) {
            gbase = atomicAdd(&sa.n[0], keep);
            if (cnt > SAMPLE_LDS) sa.n[1] = 1u; // the sample of this workgroup is truncated: the estimate is void
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < keep; i += blockDim.x)
            if (gbase + i < sa.cap) sa.list[gbase + i] = ls[i];
            else sa.n[1] = 1u;
    }
}

// distinct k-mers of the sample: every key is inserted into a scratch table (all-ones = free); a successful claim counts
__global__ void __launch_bounds__(256) k_sample_distinct(const uint64_t *list, uint32_t n, uint64_t *table, uint32_t mask,
                                                         uint32_t *n_distinct) {
    uint32_t mine = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t key = list[i];
        uint32_t off = (uint32_t) (khash(key) >> 32) & mask;
        for (uint32_t probes = 0; probes <= mask; probes++) {
            const unsigned long long old = atomicCAS((unsigned long long *) &table[off], (unsigned long long) CKEY_EMPTY, (unsigned long long) key);
            if (old == CKEY_EMPTY) { mine++; break; }
            if (old == key) break;
            off = (off + 1) & mask;
        }
    }
    if (mine) atomicAdd(n_distinct, mine);
}

// ---- finalize of the MERGE route: the entries this rank does not own leave the table ---------------------------------------
// pass 1: entries per owner; pass 2: (key, count) appended to the owner's range of the send lists, the slot's count zeroed
// (a zero count is "never seen" for every reader of the table; the key stays as a tombstone of the probe chain)
__global__ void __launch_bounds__(256) k_owner_census(CountTable t, uint64_t nslots, int w32, uint32_t n_parts,
                                                      unsigned long long *per_owner) {
    extern __shared__ uint32_t lo[]; // [n_parts] entries per owner, [n_parts]: tombstones of earlier finalizes (occupied, count zero)
    for (uint32_t b = threadIdx.x; b <= n_parts; b += blockDim.x) lo[b] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (uint64_t) gridDim.x * blockDim.x) {
        uint64_t key;
        uint32_t cnt;
        if (slot_read<true>(t, i, key, cnt)) atomicAdd(&lo[cnt != 0u ? kmer_owner(key, w32, n_parts) : n_parts], 1u); // (lo[me]: the entries that stay)
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b <= n_parts; b += blockDim.x)
        if (lo[b]) atomicAdd(&per_owner[b], (unsigned long long) lo[b]);
}
__global__ void __launch_bounds__(256) k_owner_emit(CountTable t, uint64_t nslots, int w32, uint32_t me, uint32_t n_parts,
                                                    unsigned long long *cursor /* starts of the owners' ranges */,
                                                    uint64_t *out_k, uint32_t *out_c) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    const uint64_t rounds = (nslots + stride - 1) / stride; // wave-uniform trip count (ballots inside)
    for (uint64_t it = 0; it < rounds; it++) {
        const uint64_t i = it * stride + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
        uint64_t key = CKEY_EMPTY;
        uint32_t cnt = 0, o = me;
        if (i < nslots && slot_read<true>(t, i, key, cnt)) {
            if (cnt) o = kmer_owner(key, w32, n_parts);
        }
        bool send = o != me;
        // one global atomic per wave and owner present in the wave
        uint64_t todo = __ballot(send);
        while (todo) {
            const int leader = __ffsll((unsigned long long) todo) - 1;
            const uint32_t ow = bcast_u32(o, leader);
            const uint64_t grp = __ballot(send && o == ow);
            uint64_t base = 0;
            if (lane_id() == leader) base = atomicAdd(&cursor[ow], (unsigned long long) __popcll(grp));
            base = ((uint64_t) bcast_u32((uint32_t) (base >> 32), leader) << 32) | bcast_u32((uint32_t) base, leader);
            if (send && o == ow) {
                const uint64_t pos = base + (uint64_t) __popcll(grp & ((1ull << lane_id()) - 1ull));
                out_k[pos] = key;
                out_c[pos] = cnt;
                slot_zero_count(t, i);
            }
            todo &= ~grp;
        }
    }
}

// level 1 scan, step a: one workgroup per bin -> exclusive prefix over the units + bin total
__global__ void __launch_bounds__(256) k_part_scan1a(const uint32_t *hist1, PartPlan pl, uint64_t *offs1, uint64_t *tot1) {
    __shared__ uint64_t part[256];
    const uint32_t bins1 = pl.owner_parts ? pl.owner_parts : 1u << pl.b1, b = blockIdx.x, U = pl.units1;
    const uint32_t per = (U + 255) / 256;
    const uint32_t u0 = threadIdx.x * per < U ? threadIdx.x * per : U, u1 = u0 + per < U ? u0 + per : U;
    uint64_t sum = 0;
    for (uint32_t u = u0; u < u1; u++) sum += hist1[(uint64_t) u * bins1 + b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 256; i++) { uint64_t v = part[i]; part[i] = run; run += v; }
        tot1[b] = run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t u = u0; u < u1; u++) {
        offs1[(uint64_t) u * bins1 + b] = run;
        run += hist1[(uint64_t) u * bins1 + b];
    }
}

// level 1 scan, step b: exclusive scan of the bin totals (single workgroup); binstart1[bins1] = number of k-mers
__global__ void __launch_bounds__(256) k_part_scan1b(const uint64_t *tot1, PartPlan pl, uint64_t *binstart1) {
    __shared__ uint64_t part[256];
    const uint32_t bins1 = pl.owner_parts ? pl.owner_parts : 1u << pl.b1;
    const uint32_t per = (bins1 + 255) / 256;
    const uint32_t b0 = threadIdx.x * per < bins1 ? threadIdx.x * per : bins1, b1 = b0 + per < bins1 ? b0 + per : bins1;
    uint64_t sum = 0;
    for (uint32_t b = b0; b < b1; b++) sum += tot1[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 256; i++) { uint64_t v = part[i]; part[i] = run; run += v; }
        binstart1[bins1] = run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t b = b0; b < b1; b++) { binstart1[b] = run; run += tot1[b]; }
}

// ---- LDS-staged scatter -----------------------------------------------------------------------------------
// A 1024-thread workgroup sorts a tile of <= 16384 k-mers by their digit inside LDS (rank by ds_add_rtn, in-place
// exclusive scan, 8-byte staging writes) and copies the sorted tile out, so that the k-mers of one bin leave as one
// contiguous run (full sectors) instead of isolated 8-byte stores (which cost a 32-byte HBM write each: measured
// 3.7x write amplification).  The bin of a staged k-mer is recomputed from the k-mer on the way out.
__device__ __forceinline__ void vm_wait_all() { __builtin_amdgcn_s_waitcnt(0x0F70); } // vmcnt(0), expcnt / lgkmcnt untouched
static constexpr uint32_t TILE_ITEMS = 16384;
static constexpr int SCATTER_THREADS = 1024;

struct ScatterLds {
    uint64_t *stage;  // TILE_ITEMS
    uint64_t *gbase;  // nbins: next free global position of this unit for every bin
    uint32_t *lstart; // nbins + 1: counts, then exclusive starts inside the tile
    uint32_t *wtot;   // 16 wave totals
};
__device__ __forceinline__ ScatterLds scatter_lds(uint8_t *smem, uint32_t nbins) {
    ScatterLds l;
    l.stage = reinterpret_cast<uint64_t *>(smem);
    l.gbase = l.stage + TILE_ITEMS;
    l.lstart = reinterpret_cast<uint32_t *>(l.gbase + nbins);
    l.wtot = l.lstart + nbins + 1;
    return l;
}
static size_t scatter_lds_bytes(uint32_t nbins) { return (size_t) TILE_ITEMS * 8 + (size_t) nbins * 8 + ((size_t) nbins + 1 + 16) * 4 + 16; }

// What a partition item is: IT_HASH = khash(key) (digit = a bit field of the item), IT_KEY = the key itself (digit from
// khash(key)), IT_OWNER = the key, digit = its owner in a `shift`-way key partition (DispatchableT,
// kmercount.rs:382-420; `mask` != 0 marks 32-bit k-mer values).
enum { IT_HASH = 0, IT_KEY = 1, IT_OWNER = 2, IT_KEY_TO_HASH = 3 /* k_arr_scatter: keys in, khash(key) out */ };
template <int IT>
__device__ __forceinline__ uint32_t digit_of(uint64_t item, int region_bits, int shift, uint32_t mask) {
    if (IT == IT_OWNER) return kmer_owner(item, mask != 0u, (uint32_t) shift);
    return (region_of_hash(IT == IT_HASH ? item : khash(item), region_bits) >> shift) & mask;
}

// Segmented output (the single-pass partition, see partitioned_add): bin b of this unit owns the fixed range
// [(bin_base + b) * bincap + end_rel - cap, (bin_base + b) * bincap + end_rel) of `out`, sized from the expected number of
// items with a margin; an item beyond its range is dropped and reported in *ovf (the caller then takes the exact route).
#ifndef KMU_LEAF6_WIDE // (diagnostic builds: 1 = the second plane of 6-byte leaf items holds 32-bit words)
#define KMU_LEAF6_WIDE 0
#endif
// Where the two parts of 6-byte item i of the leaf array live.  KMU_LEAF6_BLOCK = 1: in blocks of eight items, 48 bytes = eight
// u32 low words then eight u16 high parts, so that the two stores of an item (and of its neighbours in a run) land next to each
// other; 0: two planes, all low words then all high parts (leaf capacities are multiples of 16 items: a block never straddles leaves).
#ifndef KMU_LEAF6_BLOCK
#define KMU_LEAF6_BLOCK 1
#endif
__device__ __forceinline__ void leaf6_store(uint64_t *out, uint64_t n_total, uint64_t at, uint64_t v) {
#if KMU_LEAF6_BLOCK && !KMU_LEAF6_WIDE
    uint8_t *b = reinterpret_cast<uint8_t *>(out) + (at >> 3) * 48u;
    reinterpret_cast<uint32_t *>(b)[at & 7u] = (uint32_t) v;
    reinterpret_cast<uint16_t *>(b + 32)[at & 7u] = (uint16_t) (v >> 32);
#else
    reinterpret_cast<uint32_t *>(out)[at] = (uint32_t) v;
#if KMU_LEAF6_WIDE
    (reinterpret_cast<uint32_t *>(out) + n_total)[at] = (uint32_t) (v >> 32);
#else
    reinterpret_cast<uint16_t *>(reinterpret_cast<uint32_t *>(out) + n_total)[at] = (uint16_t) (v >> 32);
#endif
#endif
}
__device__ __forceinline__ uint64_t leaf6_load(const uint64_t *items, uint64_t n_total, uint64_t at) {
#if KMU_LEAF6_BLOCK && !KMU_LEAF6_WIDE
    const uint8_t *b = reinterpret_cast<const uint8_t *>(items) + (at >> 3) * 48u;
    return ((uint64_t) reinterpret_cast<const uint16_t *>(b + 32)[at & 7u] << 32) | reinterpret_cast<const uint32_t *>(b)[at & 7u];
#elif KMU_LEAF6_WIDE
    return ((uint64_t) (reinterpret_cast<const uint32_t *>(items) + n_total)[at] << 32) | reinterpret_cast<const uint32_t *>(items)[at];
#else
    return ((uint64_t) reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint32_t *>(items) + n_total)[at] << 32) | reinterpret_cast<const uint32_t *>(items)[at];
#endif
}
// CHUNKED streams (the shared level-1 streams): stream (block, bin) of `cap` items is not contiguous -- it is cut into chunks of
// 2^lc items, and chunk c of all the bins of a block lie side by side: item `rel` of the stream sits at
// ((block * cap / 2^lc + rel / 2^lc) * bins + bin) * 2^lc + rel % 2^lc.  The cursors of a block's 2 048 streams advance at the
// same pace, so what a workgroup writes at any time lies within two or three chunk rows (16 - 50 MB: a handful of pages) instead
// of 2 048 places a megabyte apart (2.2 GB: a page each).  Opt-in (KMU_COUNT_SEG_CHUNK): see seg_chunk_log.
__host__ __device__ __forceinline__ uint64_t seg_chunked_index(uint32_t block, uint32_t nbins, uint32_t bin, uint32_t cap, uint32_t rel, uint32_t lc) {
    return ((((uint64_t) block * (cap >> lc) + (rel >> lc)) * nbins + bin) << lc) | (rel & ((1u << lc) - 1u));
}
struct SegOut {
    uint64_t bin_base, bincap, end_rel, cap;
    uint32_t *ovf;
    uint32_t lc = 0, cblock = 0, cbins = 0; // lc != 0: CHUNKED -- the streams of block `cblock` of `cbins` bins (bin_base / bincap / end_rel unused)
    // LEAF6 (tile_scatter_seg): the output holds 6 bytes per item in two planes -- u32 low words of all items, then u16 bits 32..47;
    // n_total = items in all streams (where the second plane starts)
    uint64_t n_total = 0;
};

// it[j] == CKEY_EMPTY marks "no k-mer".  All 1024 threads call this together.  (The exact levels, the owner grouping of a
// distributed add, the generic array partition; the single-pass partition has its own form, tile_scatter_seg.)
template <int IT>
__device__ __forceinline__ void tile_scatter(uint64_t (&it)[16], const ScatterLds &l, uint32_t nbins, int region_bits,
                                             int shift, uint32_t mask, uint64_t *out) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    uint32_t br[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        br[j] = 0;
        if (it[j] != CKEY_EMPTY) {
            uint32_t bin = digit_of<IT>(it[j], region_bits, shift, mask);
            uint32_t rank = atomicAdd(&l.lstart[bin], 1u);
            br[j] = (bin << 16) | rank;
        }
    }
    lds_barrier();
    // in-place exclusive scan of lstart[0..nbins) (two bins per thread); lstart[nbins] = tile total
    {
        const uint32_t b0 = 2u * tid, b1 = b0 + 1;
        const uint32_t c0 = b0 < nbins ? l.lstart[b0] : 0u, c1 = b1 < nbins ? l.lstart[b1] : 0u;
        const uint32_t incl = wave_incl_scan_u32(c0 + c1);
        if (lane_id() == 63) l.wtot[tid >> 6] = incl;
        lds_barrier();
        uint32_t wpre = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t v = l.wtot[w];
            wpre += w < (tid >> 6) ? v : 0u;
        }
        const uint32_t excl = wpre + incl - (c0 + c1);
        if (b0 < nbins) l.lstart[b0] = excl;
        if (b1 < nbins) l.lstart[b1] = excl + c0;
        if (tid == nthreads - 1) l.lstart[nbins] = wpre + incl;
    }
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (it[j] != CKEY_EMPTY) l.stage[l.lstart[br[j] >> 16] + (br[j] & 0xFFFFu)] = it[j];
    lds_barrier();
    const uint32_t total = l.lstart[nbins];
    // eight positions at a time: the staged items, then their bins' bases, are requested together (one LDS round trip
    // per batch instead of two per position)
    for (uint32_t p0 = 0; p0 < total; p0 += 8u * nthreads) {
        uint64_t v[8], dst[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = p0 + (uint32_t) u * nthreads + tid;
            v[u] = p < total ? l.stage[p] : CKEY_EMPTY;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = p0 + (uint32_t) u * nthreads + tid;
            const uint32_t bin = p < total ? digit_of<IT>(v[u], region_bits, shift, mask) : 0u;
            dst[u] = l.gbase[bin] + (uint64_t) (p - l.lstart[bin]);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = p0 + (uint32_t) u * nthreads + tid;
            if (p < total) out[dst[u]] = v[u];
        }
    }
    lds_barrier();
    uint32_t cnt[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t b = 2u * tid + q;
        cnt[q] = b < nbins ? l.lstart[b + 1] - l.lstart[b] : 0u;
    }
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t b = 2u * tid + q;
        if (b < nbins) { l.gbase[b] += cnt[q]; l.lstart[b] = 0; }
    }
    if (tid == 0) l.lstart[nbins] = 0;
    lds_barrier();
}

// ---- the tile sort of the single-pass partition (round 2) -------------------------------------------------------------
// Same tile, same runs, fewer phases: four LDS barriers per tile instead of seven and one table look-up per item on the way
// out instead of two.  The running fill of a bin's segment lives in the REGISTERS of the thread that owns the bin (thread t:
// bins 2t, 2t + 1); what the write-out needs is one 32-bit word per bin, grel = fill - start of the bin inside the tile
// (mod 2^32), so that an item at tile position p goes to slot grel[bin] + p of its segment.  The rank counters are a
// separate array that the owner zeroes while it scans them, so the ranks of the next tile are taken by the waves that are
// through with this tile's write-out while the others still store (no barrier behind the write-out).
#ifndef KMU_TILE_ALLV // (A/B builds: 0 = every item under its own branch)
#define KMU_TILE_ALLV 1
#endif
#ifndef KMU_L1_ALLV // level 1 from the bases: which phases take the branch-free form (0: none)
#define KMU_L1_ALLV 0
#endif
#ifndef KMU_TILE_ALLV_N
#define KMU_TILE_ALLV_N 4
#endif
struct SegLds {
    uint64_t *stage;  // TILE_ITEMS
    uint32_t *cnt;    // nbins + 1 (+ 1 pad): ranks handed out in this tile; [nbins]: the "no k-mer" marks
    uint32_t *lstart; // nbins + 1 (+ 1 pad): exclusive starts inside the tile
    uint32_t *grel;   // nbins
    uint32_t *wtot;   // 16 wave totals
};
__device__ __forceinline__ SegLds seg_lds(uint8_t *smem, uint32_t nbins, uint32_t tile_items = TILE_ITEMS) {
    SegLds l;
    l.stage = reinterpret_cast<uint64_t *>(smem);
    l.cnt = reinterpret_cast<uint32_t *>(l.stage + tile_items);
    l.lstart = l.cnt + nbins + 2;
    l.grel = l.lstart + nbins + 2;
    l.wtot = l.grel + nbins;
    return l;
}
static size_t seg_lds_bytes(uint32_t nbins, uint32_t tile_items = TILE_ITEMS) { return (size_t) tile_items * 8 + ((size_t) nbins * 3 + 4) * 4 + 64; }

#if KMU_DIAG
__device__ unsigned long long g_diag_seg[2][8]; // [level][phase]: thread-0 clocks of the tile sort (diagnostic builds)
#endif
struct SegClk {
#if KMU_DIAG
    uint64_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t = 0;
    __device__ __forceinline__ void start() { if (threadIdx.x == 0) t = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void mark(int i) { if (threadIdx.x == 0) { const uint64_t n = __builtin_readcyclecounter(); acc[i] += n - t; t = n; } }
    __device__ __forceinline__ void flush(int level) { if (threadIdx.x == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_diag_seg[level][i], (unsigned long long) acc[i]); }
#else
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(int) {}
#endif
};

// An item that finds its segment full goes to the spill list (k-mers that occur many times -- a genome at coverage c -- make
// a bin's fill vary sqrt(c) times more than the margin of independent k-mers allows for; the list is added to the finished
// table by direct insertion, k_count_add_spill); only a full spill list raises the flag that sends the batch to the exact
// levels.  ovf: [0] flag, [1] items spilled, [2] capacity of the list, [4..5] its address.
__device__ __forceinline__ void seg_spill(uint32_t *ovf, uint64_t item) {
    const uint32_t at = atomicAdd(&ovf[1], 1u);
    if (at < ovf[2]) (*reinterpret_cast<uint64_t *const *>(ovf + 4))[at] = item;
    else ovf[0] = 1u;
}

// items are khash values (IT_HASH); nbins a multiple of 4, <= 2048; all 1024 threads call this together; cnt[] zero on the
// first call.  The digit is a bit field of the item's high word; a destination is one v_mad_u64_u32.  (A form without the
// per-item branches -- "no k-mer" marks as items of a bin of their own behind the others -- needs 40 more registers than
// the 128 a thread has: the compiler keeps both tiles' items and all sixteen addresses live.  Tried again in round 3 with
// the LDS round trips capped at four in flight: 16 / 31 spilled registers, level 1 23.4 ms against 19.3, level 2 25.3 against 22.9.)
// VMWAIT: the caller prefetches the next tile with unconditional loads (see flat_step_fetch)
// cursor != nullptr: the segments are SHARED by the workgroups of an XCD -- a tile's run of a bin is placed by an atomic add on
// the bin's cursor (cursor[bin]: items handed out so far) instead of the unit's own fill `run`: the runs of the XCD's 32
// workgroups lie one behind the other in ONE stream per bin, so the half-written 128-byte lines at the head of a stream are
// completed by the neighbours within a tile's time instead of waiting in L2 for this workgroup's next tile (32 workgroups x
// 2 048 private streams x 128 bytes = 8 MB of open lines per XCD against 4 MB of L2: the two speeds of level 1, section 3.4).
// (Round 3, measured and not kept: level 1 taking an item's rank as soon as its hash is made, so that a wave's LDS atomics run
// under the arithmetic of its next items -- 16.2 / 17.9 / 18.0 ms against 15.7 / 16.2, scripts/r03_prerank.sh.)
// LEAF6: what leaves is bits 47..0 of an item, in two planes (SegOut::n_total): the region build needs the 64 - w hash bits
// below the region index and knows the rest from where it reads (tables of >= 2^28 slots: w >= 16); 26 instead of 35 GB written
// by level 2 and read by the build at the bench size.
template <bool VMWAIT, bool CUR = false, int THREADS = SCATTER_THREADS, bool LEAF6 = false, int ALLV = 0>
__device__ __forceinline__ void tile_scatter_seg(uint64_t (&it)[16], const SegLds &l, uint32_t nbins, int region_bits, int shift,
                                                 uint64_t *out, const SegOut &sg, uint32_t (&run)[2], SegClk &clk,
                                                 uint32_t *cursor = nullptr) {
    const uint32_t tid = threadIdx.x, nthreads = THREADS, mask = nbins - 1;
    const uint32_t sh32 = (uint32_t) (32 - region_bits + shift); // digit = (high word >> sh32) & mask  (region_bits <= 22)
    auto bin_of = [&](uint64_t item) -> uint32_t { return ((uint32_t) (item >> 32) >> sh32) & mask; };
    clk.mark(0); // everything between two tiles: the loads, the front end of level 1
    uint32_t rk[8]; // ranks (< 16384), two to a register
#pragma unroll
    for (int j = 0; j < 8; j++) rk[j] = 0;
    // A wave whose sixteen items per lane are all k-mers (nearly every wave of long reads and of the inner levels) takes its ranks
    // and stages its items without the per-item branches: the LDS requests of a lane leave back to back and are waited for once,
    // not one `s_waitcnt` per item inside sixteen EXEC regions.
    bool allv_lane = true;
#pragma unroll
    for (int j = 0; j < 16; j++) allv_lane = allv_lane && it[j] != CKEY_EMPTY;
    // (ALLV: the callers whose registers have the room -- level 1 from the bases spills 25 with it and takes 18.7 instead of 12.4 ms;
    //  the array levels: 17.1 -> 16.7 ms on the bench's level 2)
    const bool allv = ALLV && KMU_TILE_ALLV && __all(allv_lane); // ALLV: bit 0 = the ranks, bit 1 = the staging
    if ((ALLV & 1) && allv) {
        constexpr int G = KMU_TILE_ALLV_N; // ranks in flight per lane (8: level 1 spills 25 registers)
#pragma unroll
        for (int h = 0; h < 16 / G; h++) {
            uint32_t r[G];
#pragma unroll
            for (int j = 0; j < G; j++) r[j] = atomicAdd(&l.cnt[bin_of(it[G * h + j])], 1u);
#pragma unroll
            for (int j = 0; j < G / 2; j++) rk[G / 2 * h + j] = r[2 * j] | (r[2 * j + 1] << 16);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (it[j] != CKEY_EMPTY) rk[j >> 1] |= atomicAdd(&l.cnt[bin_of(it[j])], 1u) << (16 * (j & 1));
    }
    lds_barrier(); // (also: every wave is through with the last tile's write-out: stage / lstart / grel are free)
    clk.mark(1);
    const uint32_t b0 = 2u * tid;
    uint32_t c0 = 0, c1 = 0;
    if (b0 < nbins) {
        const uint2 c = *reinterpret_cast<const uint2 *>(&l.cnt[b0]);
        c0 = c.x;
        c1 = c.y;
        *reinterpret_cast<uint2 *>(&l.cnt[b0]) = make_uint2(0u, 0u);
        if (CUR) { // (the answers are looked at behind the staging)
            run[0] = atomicAdd(&cursor[b0], c0);
            run[1] = atomicAdd(&cursor[b0 + 1], c1);
        }
    }
    const uint32_t incl = wave_incl_scan_u32(c0 + c1);
    if (lane_id() == 63) l.wtot[tid >> 6] = incl;
    lds_barrier();
    uint32_t wpre = 0, total = 0; // total: the k-mers of the tile
    {
        const uint4 *w4 = reinterpret_cast<const uint4 *>(l.wtot);
        const uint32_t wave = tid >> 6;
#pragma unroll
        for (int q = 0; q < THREADS / 256; q++) {
            const uint4 v = w4[q];
            const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int z = 0; z < 4; z++) {
                wpre += (uint32_t) (q * 4 + z) < wave ? e[z] : 0u;
                total += e[z];
            }
        }
    }
    if (b0 < nbins) {
        const uint32_t excl = wpre + incl - (c0 + c1);
        *reinterpret_cast<uint2 *>(&l.lstart[b0]) = make_uint2(excl, excl + c0);
        if (!CUR) {
            *reinterpret_cast<uint2 *>(&l.grel[b0]) = make_uint2(run[0] - excl, run[1] - (excl + c0));
            run[0] += c0;
            run[1] += c1;
        }
    }
    lds_barrier();
    clk.mark(2);
    if ((ALLV & 2) && allv) {
        constexpr int G = KMU_TILE_ALLV_N;
#pragma unroll
        for (int h = 0; h < 16 / G; h++) {
            uint32_t at[G];
#pragma unroll
            for (int j = 0; j < G; j++) at[j] = l.lstart[bin_of(it[G * h + j])];
#pragma unroll
            for (int j = 0; j < G; j++) l.stage[at[j] + ((rk[(G * h + j) >> 1] >> (16 * (j & 1))) & 0xFFFFu)] = it[G * h + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (it[j] != CKEY_EMPTY) l.stage[l.lstart[bin_of(it[j])] + ((rk[j >> 1] >> (16 * (j & 1))) & 0xFFFFu)] = it[j];
    }
    if (CUR && b0 < nbins) {
        const uint2 ls = *reinterpret_cast<const uint2 *>(&l.lstart[b0]);
        *reinterpret_cast<uint2 *>(&l.grel[b0]) = make_uint2(run[0] - ls.x, run[1] - ls.y);
    }
    lds_barrier();
    clk.mark(3);
    if (VMWAIT) vm_wait_all(); // the next tile's requests (in flight since before the ranks) and the last tile's stores: nothing younger
    const uint64_t seg0 = sg.end_rel - sg.cap; // start of this unit's segment inside a bin's range
    const uint32_t bb = (uint32_t) sg.bin_base, bc = (uint32_t) sg.bincap, cap = (uint32_t) sg.cap;
    const uint32_t clc = LEAF6 ? 0u : (uint32_t) __builtin_amdgcn_readfirstlane((int) sg.lc); // (uniform)
    for (uint32_t p0 = 0; p0 < total; p0 += 8u * nthreads) {
        uint64_t v[8];
        uint32_t rel[8], bin[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = p0 + (uint32_t) u * nthreads + tid;
            v[u] = l.stage[p < total ? p : 0u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = p0 + (uint32_t) u * nthreads + tid;
            bin[u] = ((uint32_t) (v[u] >> 32) >> sh32) & mask;
            rel[u] = l.grel[bin[u]] + p;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = p0 + (uint32_t) u * nthreads + tid;
            if (p < total) {
#if KMU_SCATTER_NT // (A/B builds: non-temporal stores for the partition streams)
                if (rel[u] < cap) __builtin_nontemporal_store(v[u], &out[(uint64_t) (bb + bin[u]) * bc + (seg0 + rel[u])]);
#else
                if (LEAF6 && rel[u] < cap) leaf6_store(out, sg.n_total, (uint64_t) (bb + bin[u]) * bc + (seg0 + rel[u]), v[u]);
                else if (rel[u] < cap) out[clc ? seg_chunked_index(sg.cblock, sg.cbins, bin[u], cap, rel[u], clc) : (uint64_t) (bb + bin[u]) * bc + (seg0 + rel[u])] = v[u];
#endif
                else seg_spill(sg.ovf, v[u]);
            }
        }
    }
    clk.mark(4);
}

// after the last tile.  fill = true: the unused tail of every segment of this unit gets "no k-mer" marks (the next level
// reads whole bins); wave w takes bins w, w + 16, ..., its lanes store consecutive items.  fill = false: the fills go to
// counts[bin_base + b] instead (a leaf has one writer: the build reads exactly the items that are there).
__device__ __forceinline__ void seg_finish_unit(const SegLds &l, uint32_t nbins, const SegOut &sg, const uint32_t (&run)[2], uint64_t *out,
                                                bool fill, uint32_t *counts) {
    const uint32_t tid = threadIdx.x, b0 = 2u * tid;
    const uint32_t f0 = (uint64_t) run[0] < sg.cap ? run[0] : (uint32_t) sg.cap, f1 = (uint64_t) run[1] < sg.cap ? run[1] : (uint32_t) sg.cap;
    if (!fill) {
        if (b0 < nbins) *reinterpret_cast<uint2 *>(&counts[sg.bin_base + b0]) = make_uint2(f0, f1);
        return;
    }
    lds_barrier(); // (the last write-out has read grel / stage; lstart is free in any case)
    if (b0 < nbins) *reinterpret_cast<uint2 *>(&l.lstart[b0]) = make_uint2(f0, f1);
    lds_barrier();
    const uint32_t wave = tid >> 6, nwaves = SCATTER_THREADS >> 6, lane = (uint32_t) lane_id();
    const uint64_t seg0 = sg.end_rel - sg.cap;
    for (uint32_t b = wave; b < nbins; b += nwaves) {
        const uint64_t base = (sg.bin_base + b) * sg.bincap + seg0;
        const uint32_t cur = (uint32_t) __builtin_amdgcn_readfirstlane((int) l.lstart[b]);
        for (uint64_t i = cur + lane; i < sg.cap; i += 64) out[base + i] = CKEY_EMPTY;
    }
}

// the code words of wave step `st`: this lane's word and (lanes 0/1) the two words after the wave's last
__device__ __forceinline__ void flat_step_load(const uint8_t *bases, uint64_t total, uint64_t st, bool active, uint32_t &w0,
                                               uint32_t &ex, uint32_t *bad_acc = nullptr) {
    w0 = 0;
    ex = 0;
    if (!active) return; // wave-uniform
    SeqView s;
    s.base = bases; s.begin = 0; s.len = total; s.total = total; s.packed = 0;
    uint32_t bad, bad2;
    w0 = load_code_word(s, st * 64 + (uint64_t) lane_id(), bad);
    ex = load_code_word(s, st * 64 + 64 + (uint64_t) (lane_id() & 1), bad2);
    if (bad_acc) *bad_acc |= bad; // (every word is some step's own word: the halo words need no second look)
}

// The same in two halves, for prefetching: flat_step_fetch requests the two aligned 16-byte chunks (this lane's word, the
// halo word of lane & 1) and nothing looks at them until flat_step_words turns them into code words one tile later -- the
// requests are UNCONDITIONAL loads from clamped addresses (needs total >= 16), so that their number in flight is a constant
// for the compiler's s_waitcnt placement (a load under a branch makes it wait for everything at the first use of anything).
// All vector-memory waits of the scatter loops are the explicit vmcnt(0) of vm_wait_all(): once before the loop, once per
// tile just before the write-out, when the requests of the next tile have had the whole tile sort to arrive and the stores
// of the last tile are long gone.
struct FlatRaw {
    uint4 c0, cx;
    uint4 tb; // the step's entry of the read table (k_step_table), requested with the chunks
    uint32_t nv; // KMU_STEP_TAB == 2: the lane's 16 "no k-mer" bits (flat_novalid)
};
// The read that holds the first base of every wave step, and where it and the next two reads end, made once per call: x = the
// read's index, y / z / w = the ends relative to the step's first base (0xFFFFFFFF: more than 4 G bases away).  Level 1 used to
// look these up inside the step -- offsets[hint], offsets[hint + 1 + lane], offsets[r + 1], and offsets[r + 2] ... in the lanes
// behind a read end: dependent loads that the other fifteen waves of the tile wait for at the next barrier (a tile of sixteen
// steps of 1 024 bases meets a read end more often than not); an entry is requested a tile ahead, next to the step's bases.
__global__ void __launch_bounds__(256) k_step_table(const uint64_t *offsets, uint32_t n_seq, uint64_t nsteps, uint4 *tab) {
    const uint64_t st = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (st >= nsteps) return;
    const uint64_t total = offsets[n_seq];
    const uint64_t g = st * 1024 < total ? st * 1024 : (total ? total - 1 : 0);
    uint32_t lo = 0, hi = n_seq; // largest i with offsets[i] <= g (0 if there is none): wave_find_read's answer
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (offsets[mid] <= g) lo = mid;
        else hi = mid;
    }
    const uint64_t s0 = st * 1024;
    uint32_t rel[3];
    for (int i = 0; i < 3; i++) { // the ends of reads lo, lo + 1, lo + 2 (a step and its halo rarely see more)
        const uint64_t e = offsets[(uint64_t) lo + 1 + i < n_seq ? lo + 1 + i : n_seq];
        rel[i] = e > s0 ? (e - s0 < 0xFFFFFFFFull ? (uint32_t) (e - s0) : 0xFFFFFFFFu) : 0u;
    }
    tab[st] = make_uint4(lo, rel[0], rel[1], rel[2]);
}
template <bool TAB = false>
__device__ __forceinline__ void flat_step_fetch(const uint8_t *bases, uint64_t total, uint64_t st, FlatRaw &r, const uint4 *tab = nullptr,
                                                uint64_t last_step = 0) {
    r.c0 = make_uint4(0u, 0u, 0u, 0u);
    r.cx = r.c0;
#if KMU_STEP_TAB == 2
    if (TAB) r.nv = reinterpret_cast<const uint16_t *>(tab)[(st < last_step ? st : last_step) * 64 + (uint64_t) lane_id()];
#else
    if (TAB) r.tb = tab[st < last_step ? st : last_step];
#endif
    if (total < 16) return; // (wave-uniform; flat_step_words then reads the ragged chunk itself)
    const uint64_t lastc = (total - 16) & ~15ull;
    const uint64_t a0 = (st * 64 + (uint64_t) lane_id()) * 16, ax = (st * 64 + 64 + (uint64_t) (lane_id() & 1)) * 16;
#if KMU_SCATTER_NTLOAD >= 2
    typedef uint32_t u32x4nt __attribute__((ext_vector_type(4)));
    const u32x4nt c0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4nt *>(bases + (a0 < lastc ? a0 : lastc)));
    const u32x4nt cx = __builtin_nontemporal_load(reinterpret_cast<const u32x4nt *>(bases + (ax < lastc ? ax : lastc)));
    r.c0 = make_uint4(c0.x, c0.y, c0.z, c0.w);
    r.cx = make_uint4(cx.x, cx.y, cx.z, cx.w);
#else
    r.c0 = *reinterpret_cast<const uint4 *>(bases + (a0 < lastc ? a0 : lastc));
    r.cx = *reinterpret_cast<const uint4 *>(bases + (ax < lastc ? ax : lastc));
#endif
}
__device__ __forceinline__ void flat_step_words(const uint8_t *bases, uint64_t total, uint64_t st, bool active, const FlatRaw &r,
                                                uint32_t &w0, uint32_t &ex, uint32_t &bad_acc) {
    w0 = 0;
    ex = 0;
    if (!active) return; // wave-uniform
    const uint64_t i0 = st * 64 + (uint64_t) lane_id(), ix = st * 64 + 64 + (uint64_t) (lane_id() & 1);
    uint32_t bad = 0, bad2 = 0;
    if (__all(ix * 16 + 16 <= total)) { // (every chunk of the step whole: all but the last step of the stream)
        w0 = pack16_ascii(r.c0, bad);
        ex = pack16_ascii(r.cx, bad2);
    } else {
        SeqView s;
        s.base = bases; s.begin = 0; s.len = total; s.total = total; s.packed = 0;
        w0 = load_code_word(s, i0, bad);
        ex = load_code_word(s, ix, bad2);
    }
    bad_acc |= bad; // (every word is some step's own word: the halo words need no second look)
}

// up to 16 canonical k-mers of this lane for wave step `st` (CKEY_EMPTY where a k-mer would straddle a read end)
template <bool TAB = false>
__device__ __forceinline__ void flat_step_items(const uint64_t *offsets, uint32_t n_seq, uint64_t total, uint64_t start, int k,
                                                uint64_t st, bool active, uint32_t w0, uint32_t ex, uint32_t &r_hint,
                                                uint64_t (&it)[16], uint4 tb = make_uint4(0u, 0u, 0u, 0u)) {
#pragma unroll
    for (int j = 0; j < 16; j++) it[j] = CKEY_EMPTY;
    if (!active) return; // wave-uniform
    const int lane = lane_id();
    const uint64_t widx = st * 64 + lane;
    uint32_t e0 = bcast_u32(ex, 0), e1 = bcast_u32(ex, 1);
    uint32_t w1 = shfl_down_u32(w0, 1), w2 = shfl_down_u32(w0, 2);
    if (lane == 63) { w1 = e0; w2 = e1; }
    if (lane == 62) { w2 = e0; }
    uint32_t r;
    const uint64_t g0 = widx * 16;
    const bool in = g0 < total && g0 + 16 > start;
    uint64_t rend = 0;
    uint32_t r0 = 0;
    uint64_t te0 = 0, te1 = 0, te2 = 0; // TAB: the wave-uniform entry of the step (k_step_table)
    // the end of read rr (>= r0): from the entry for the first three, a load beyond
    auto end_of = [&](uint32_t rr) -> uint64_t {
        const uint32_t d = rr - r0;
        return d == 0u ? te0 : d == 1u ? te1 : d == 2u ? te2 : offsets[rr + 1];
    };
    if (TAB) {
        auto abs_end = [&](uint32_t rel) -> uint64_t {
            const uint32_t u = (uint32_t) __builtin_amdgcn_readfirstlane((int) rel);
            return u == 0xFFFFFFFFu ? ~0ull : st * 1024 + u;
        };
        r0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) tb.x);
        te0 = abs_end(tb.y);
        te1 = abs_end(tb.z);
        te2 = abs_end(tb.w);
        r = r0;
        rend = te0;
        if (in)
            while (g0 >= rend && r + 1 < n_seq) { r++; rend = end_of(r); } // the read of this lane's first base
    } else {
        r = wave_find_read_from(offsets, n_seq, st * 1024 < total ? st * 1024 : total - 1, r_hint);
        r_hint = r;
        if (in) {
            rend = offsets[r + 1];
            while (g0 >= rend && r + 1 < n_seq) { r++; rend = offsets[r + 1]; } // the read of this lane's first base
        }
    }
    const uint64_t hi = ((uint64_t) w0 << 32) | w1; // (this form keeps the reverse complement per k-mer: the exact levels' kernel has no registers for the window's)
    const int sh = 64 - 2 * k;
    if (__all(!in || (g0 >= start && rend - g0 >= (uint64_t) (15 + k)))) { // every lane well inside a read: no per-k-mer boundary tests
        if (in) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint64_t val = ((hi << (2 * j)) | (((uint64_t) w2 << (2 * j)) >> 32)) >> sh, rc = revcomp_val(val, k);
                it[j] = rc < val ? rc : val;
            }
        }
    } else if (in) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint64_t g = g0 + j;
            while (g >= rend && r + 1 < n_seq) { r++; rend = TAB ? end_of(r) : offsets[r + 1]; }
            if (g >= start && g + k <= rend) {
                const uint64_t val = ((hi << (2 * j)) | (((uint64_t) w2 << (2 * j)) >> 32)) >> sh, rc = revcomp_val(val, k);
                it[j] = rc < val ? rc : val;
            }
        }
    }
}

// the same from the lane's "no k-mer" bits: no read offsets, no search, no dependent look-up; a wave whose lanes are all-or-nothing
// (long reads: nearly every wave) skips the per-k-mer tests
__device__ __forceinline__ void flat_step_items_nv(int k, bool active, uint32_t w0, uint32_t ex, uint32_t nv, uint64_t (&it)[16]) {
#pragma unroll
    for (int j = 0; j < 16; j++) it[j] = CKEY_EMPTY;
    if (!active) return; // wave-uniform
    const int lane = lane_id();
    uint32_t e0 = bcast_u32(ex, 0), e1 = bcast_u32(ex, 1);
    uint32_t w1 = shfl_down_u32(w0, 1), w2 = shfl_down_u32(w0, 2);
    if (lane == 63) { w1 = e0; w2 = e1; }
    if (lane == 62) { w2 = e0; }
    const uint32_t V = ~nv & 0xFFFFu;
    const StepWin sw = step_win(w0, w1, w2, k);
    if (__all(V == 0xFFFFu || V == 0u)) {
        if (V) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                it[j] = step_canonical(sw, j);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if ((V >> j) & 1u) {
                it[j] = step_canonical(sw, j);
            }
        }
    }
}

// level 1, pass 2: scatter the canonical k-mers into their level-1 partitions (private ranges per unit)
// seg.cap != 0: the single-pass form -- no histogram ran; unit u (= seg.unit_base + blockIdx.x) writes bin b into its own
// segment [b * seg.bincap + u * seg.cap, + seg.cap) of `out` (offs1 / binstart1 are not read), and validates the bases.
struct SegPlan1 {
    uint64_t cap, bincap, unit_base, step_base; // step_base: first wave step of blockIdx.x == 0 (chunked calls)
    uint32_t *ovf;
    uint32_t *err;
    // rounds (kmu_sketch_count under an upload): the same units take a slice of every round's wave steps [step_base, step_end)
    // and keep the fills of their segments in `state` ([unit][bin]) between the launches; the tails are marked in the last one
    uint64_t step_end;
    uint32_t *state;
    int first, last;
    // sets != 0: shared segments -- `state` holds one cursor per (set, bin), set = blockIdx.x % sets (the workgroups of an XCD:
    // the dispatcher deals them out round robin), cap items per (set, bin) stream at out[(set * bins + bin) * cap]; nothing is
    // loaded, stored or marked by the kernel (k_seg_tails marks the tails behind the last launch)
    uint32_t sets;
    const uint4 *step_tab; // k_step_table's entries of all wave steps of the stream (single-pass form)
    uint32_t lc = 0;       // sets != 0: the streams are CHUNKED (SegOut)
};
template <bool SEGM, bool SHARED = false>
__global__ void __launch_bounds__(1024) k_part_scatter1(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq,
                                                        int k, PartPlan pl, const uint64_t *offs1,
                                                        const uint64_t *binstart1, uint64_t *out, SegPlan1 seg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t bins1 = pl.owner_parts ? pl.owner_parts : 1u << pl.b1;
    ScatterLds l = scatter_lds(smem, bins1);
    SegLds ls = seg_lds(smem, bins1);
    const uint64_t unit = seg.unit_base + blockIdx.x;
    // single-pass form: the segments of a unit lie side by side (block `unit` of bins1 segments): a workgroup's 2^b1 output
    // streams stay inside bins1 x cap items (148 MB at the bench size) instead of spreading over the whole buffer, one
    // stream every 18 MB (18.6 against 20.1 ms on the bench workload, same box, alternating processes; the kernel also has
    // a slow state of the box -- 25 ms with either layout for minutes on end -- that no layout changes)
    SegOut sg{unit * bins1, seg.cap, seg.cap, seg.cap, seg.ovf};
    if (seg.bincap) sg = SegOut{0, seg.bincap, (unit + 1) * seg.cap, seg.cap, seg.ovf}; // (A/B: the bins' ranges side by side, KMU_COUNT_SEG_LAYOUT=bin)
    uint32_t *cursor = nullptr;
    if (SEGM && SHARED) {
        const uint32_t set = seg.first & 2 ? blockIdx.x / ((gridDim.x + seg.sets - 1) / seg.sets) : blockIdx.x % seg.sets; // (A/B: bit 1 of first = sets of consecutive workgroups)
        sg = SegOut{(uint64_t) set * bins1, seg.cap, seg.cap, seg.cap, seg.ovf};
        sg.lc = seg.lc; sg.cblock = set; sg.cbins = bins1;
        cursor = seg.state + (size_t) set * bins1;
    }
    uint32_t run[2] = {0u, 0u};
    SegClk clk;
    if (SEGM) {
        for (uint32_t b = threadIdx.x; b <= bins1; b += blockDim.x) ls.cnt[b] = 0;
    } else {
        for (uint32_t b = threadIdx.x; b < bins1; b += blockDim.x) {
            l.gbase[b] = binstart1[b] + offs1[(uint64_t) blockIdx.x * bins1 + b];
            l.lstart[b] = 0;
        }
        if (threadIdx.x == 0) l.lstart[bins1] = 0;
    }
    lds_barrier();
    const uint64_t total = offsets[n_seq], start = offsets[0];
    const uint64_t nsteps_all = ((total + 15) / 16 + 63) / 64;
    const uint64_t nsteps = SEGM && seg.step_end ? seg.step_end : nsteps_all;
    const uint64_t s0 = seg.step_base + (uint64_t) blockIdx.x * pl.steps_per_unit;
    const uint64_t s1 = s0 + pl.steps_per_unit < nsteps ? s0 + pl.steps_per_unit : nsteps;
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t r_hint = 0xFFFFFFFFu, w0, ex, bad = 0;
    FlatRaw raw;
    if (SEGM && !SHARED && seg.state && !(seg.first & 1) && 2u * threadIdx.x < bins1) { // the fills of this unit's segments so far
        const uint2 f = *reinterpret_cast<const uint2 *>(&seg.state[unit * bins1 + 2u * threadIdx.x]);
        run[0] = f.x;
        run[1] = f.y;
    }
    const uint64_t last_step = nsteps_all ? nsteps_all - 1 : 0;
    if (SEGM) {
        flat_step_fetch<KMU_STEP_TAB != 0>(bases, total, s0 + wave, raw, seg.step_tab, last_step);
        vm_wait_all();
    }
    clk.start();
    for (uint64_t t0 = s0; t0 < s1; t0 += nwaves) {
        uint64_t it[16];
        if (SEGM) flat_step_words(bases, total, t0 + wave, t0 + wave < s1, raw, w0, ex, bad);
        else flat_step_load(bases, total, t0 + wave, t0 + wave < s1, w0, ex, &bad); // (the exact levels and the owner grouping: the older tile sort leaves no registers for a chunk in flight)
#if KMU_STEP_TAB == 2
        if (SEGM) flat_step_items_nv(k, t0 + wave < s1, w0, ex, raw.nv, it);
#else
        if (SEGM) flat_step_items<KMU_STEP_TAB != 0>(offsets, n_seq, total, start, k, t0 + wave, t0 + wave < s1, w0, ex, r_hint, it, raw.tb);
#endif
        else flat_step_items(offsets, n_seq, total, start, k, t0 + wave, t0 + wave < s1, w0, ex, r_hint, it);
        // the next step's chunks and its entry of the read table are requested now; they arrive under the tile sort, which waits
        // for them before its write-out
        if (SEGM) flat_step_fetch<KMU_STEP_TAB != 0>(bases, total, t0 + nwaves + wave, raw, seg.step_tab, last_step);
        if (!SEGM && pl.owner_parts) tile_scatter<IT_OWNER>(it, l, bins1, -1, (int) pl.owner_parts, (uint32_t) pl.owner_w32, out);
        else { // from here on the k-mers travel as their table hash
#pragma unroll
            for (int j = 0; j < 16; j++) it[j] = khash(it[j]); // (khash keeps the "no k-mer" mark)
            if (SEGM) tile_scatter_seg<true, SHARED, SCATTER_THREADS, false, KMU_L1_ALLV>(it, ls, bins1, pl.region_bits, pl.b2, out, sg, run, clk, cursor);
            else tile_scatter<IT_HASH>(it, l, bins1, pl.region_bits, pl.b2, bins1 - 1, out);
        }
    }
    if (SEGM) {
        if (SHARED) {
        } else if (seg.state && !seg.last) {
            if (2u * threadIdx.x < bins1) *reinterpret_cast<uint2 *>(&seg.state[unit * bins1 + 2u * threadIdx.x]) = make_uint2(run[0], run[1]);
        } else seg_finish_unit(ls, bins1, sg, run, out, true, nullptr);
        clk.mark(5);
        clk.flush(0);
        if (bad) atomicOr(seg.err, DERR_NON_ACGT); // (the histogram pass that used to validate the bases did not run)
    }
}

// launch the two scan steps (host side)
static inline uint32_t scan_threads_per_bin(uint32_t chunks) {
    uint32_t T = 1;
    while (T * 2 <= chunks && T < 256) T *= 2;
    return T;
}

// ---- generic radix partition of a u64 array (level 2 of the read path; both levels of the array path) ------------
// The input is a set of `nparts` consecutive partitions (bounds[nparts + 1]); every partition is cut into `chunks`
// units; a unit scatters its slice by the digit (region >> shift) & (bins - 1) into bins sub-partitions.
struct ArrPlan {
    int region_bits;
    int shift;
    uint32_t bins;
    uint32_t nparts;
    uint32_t chunks;
    // != 0: the input partition p is not contiguous but the p-th segment (seg_cap items) of each of seg_units blocks of
    // seg_bins segments -- what the single-pass level 1 leaves, one block per unit (k_part_scatter1<true>); bounds is not read
    uint32_t seg_units, seg_cap, seg_bins;
    // k_arr_scatter<.., SHARED>: > 1 = that many sets of shared output streams per partition, a unit writes set blockIdx.x % out_sets
    // (level 1 of an array: [set][bin][seg_cap], cursors in the same order); 0 / 1: one set (level 2: the leaves)
    uint32_t out_sets;
    uint32_t seg_lc = 0; // the input streams are CHUNKED (SegOut)
    uint32_t out_lc = 0; // out_sets > 1: so are the output streams
};

__device__ __forceinline__ void arr_unit_range(const uint64_t *bounds, const ArrPlan &pl, uint32_t unit, uint64_t *i0,
                                               uint64_t *i1) {
    const uint32_t part = unit / pl.chunks, c = unit % pl.chunks;
    const uint64_t s = bounds[part], len = bounds[part + 1] - s;
    *i0 = s + len * c / pl.chunks;
    *i1 = s + len * (c + 1) / pl.chunks;
}

template <int IT>
__global__ void __launch_bounds__(256) k_arr_hist(const uint64_t *in, const uint64_t *bounds, ArrPlan pl, uint32_t *hist) {
    extern __shared__ uint32_t lh[];
    for (uint32_t b = threadIdx.x; b < pl.bins; b += blockDim.x) lh[b] = 0;
    __syncthreads();
    uint64_t i0, i1;
    arr_unit_range(bounds, pl, blockIdx.x, &i0, &i1);
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x)
        atomicAdd(&lh[digit_of<IT>(in[i], pl.region_bits, pl.shift, pl.bins - 1)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < pl.bins; b += blockDim.x) hist[(uint64_t) blockIdx.x * pl.bins + b] = lh[b];
}

// Offsets of the units' private output ranges; order inside a partition = (bin major, chunk minor).
// Step a: T threads share one (partition, bin): exclusive prefix of the bin's counts over the partition's chunks
// (relative offsets) and the bin total.  T = min(256, chunks) rounded down to a power of two, 256 / T bins per workgroup.
__global__ void __launch_bounds__(256) k_arr_scan_a(const uint32_t *hist, ArrPlan pl, uint32_t T, uint64_t *offs_rel,
                                                    uint64_t *tot) {
    __shared__ uint64_t part[256];
    const uint32_t bins = pl.bins, C = pl.chunks, per_wg = 256u / T;
    const uint32_t groups = (bins + per_wg - 1) / per_wg; // workgroups per partition
    const uint32_t p1 = blockIdx.x / groups, b = (blockIdx.x % groups) * per_wg + threadIdx.x / T, tc = threadIdx.x % T;
    const uint32_t per = (C + T - 1) / T;
    const uint32_t c0 = tc * per < C ? tc * per : C, c1 = c0 + per < C ? c0 + per : C;
    uint64_t sum = 0;
    if (b < bins)
        for (uint32_t c = c0; c < c1; c++) sum += hist[((uint64_t) p1 * C + c) * bins + b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (tc == 0) { // exclusive scan of this bin's T partial sums
        uint64_t run = 0;
        for (uint32_t i = 0; i < T; i++) { const uint64_t v = part[threadIdx.x + i]; part[threadIdx.x + i] = run; run += v; }
        if (b < bins) tot[(uint64_t) p1 * bins + b] = run;
    }
    __syncthreads();
    if (b < bins) {
        uint64_t run = part[threadIdx.x];
        for (uint32_t c = c0; c < c1; c++) {
            offs_rel[((uint64_t) p1 * C + c) * bins + b] = run;
            run += hist[((uint64_t) p1 * C + c) * bins + b];
        }
    }
}

// Step b: one workgroup per partition: exclusive scan of the bin totals, shifted by the partition's start ->
// outbounds[p1 * bins + b]; outbounds[nparts * bins] = end of the last partition.
__global__ void __launch_bounds__(256) k_arr_scan_b(const uint64_t *tot, const uint64_t *bounds, ArrPlan pl, uint64_t *outbounds) {
    __shared__ uint64_t part[256];
    const uint32_t bins = pl.bins, p1 = blockIdx.x;
    const uint32_t per = (bins + 255) / 256;
    const uint32_t b0 = threadIdx.x * per < bins ? threadIdx.x * per : bins, b1 = b0 + per < bins ? b0 + per : bins;
    uint64_t sum = 0;
    for (uint32_t b = b0; b < b1; b++) sum += tot[(uint64_t) p1 * bins + b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = bounds[p1];
        for (int i = 0; i < 256; i++) { uint64_t v = part[i]; part[i] = run; run += v; }
        if (p1 == pl.nparts - 1) outbounds[(uint64_t) pl.nparts * bins] = bounds[pl.nparts];
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t b = b0; b < b1; b++) {
        outbounds[(uint64_t) p1 * bins + b] = run;
        run += tot[(uint64_t) p1 * bins + b];
    }
}

// seg_cap != 0: the single-pass form -- no histogram ran; unit (partition p, chunk c) writes bin b into its own segment
// [((p * bins + b) * chunks + c) * seg_cap, + seg_cap) of `out` (offs_rel / outbounds are not read); "no k-mer" marks in the
// input (the tails of the previous level's segments) are skipped like everywhere else.
// SHARED (level 2 of the read path): the `chunks` units of an input partition write ONE set of leaves, a tile's run of a leaf
// placed by an atomic add on the leaf's cursor (leafcnt[leaf], zero before the launch; it ends as the leaf's fill -- or more,
// where items went to the spill list: the build clamps it).  The units of a partition are the workgroups 8 apart in the grid:
// the dispatcher deals workgroups out to the 8 XCDs round robin, so they run at the same time on the same XCD and its L2 sees
// their runs of a leaf side by side (see tile_scatter_seg).
template <int IT, bool SEGM, bool SHARED = false, int THREADS = SCATTER_THREADS, bool LEAF6 = false>
__global__ void __launch_bounds__(THREADS) k_arr_scatter(const uint64_t *in, const uint64_t *bounds, ArrPlan pl,
                                                      const uint64_t *offs_rel, const uint64_t *outbounds, uint64_t *out,
                                                      uint64_t seg_cap, uint32_t *seg_ovf, uint32_t *leafcnt) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t TILE = 16u * THREADS; // (THREADS < 1024: SEGM only -- half tiles, two workgroups per CU)
    ScatterLds l = scatter_lds(smem, pl.bins);
    SegLds ls = seg_lds(smem, pl.bins, TILE);
    uint64_t sp = blockIdx.x / pl.chunks;
    uint32_t my_chunk = blockIdx.x % pl.chunks;
    if (SHARED && pl.seg_units && (pl.nparts & 7u) == 0u) {
        sp = 8u * (blockIdx.x / (8u * pl.chunks)) + (blockIdx.x & 7u);
        my_chunk = (blockIdx.x >> 3) % pl.chunks;
    }
    // segmented output: the bins' segments of a unit side by side, block blockIdx.x of pl.bins segments (level 2, one unit per
    // input partition: the region leaves in region order)
    SegOut sg{(uint64_t) blockIdx.x * pl.bins, seg_cap, seg_cap, seg_cap, seg_ovf};
    uint32_t *cursor = nullptr;
    if (SHARED) {
        const uint32_t nsets = pl.out_sets > 1u ? pl.out_sets : 1u;
        const uint64_t block = sp * nsets + (nsets > 1u ? blockIdx.x % nsets : 0u);
        sg = SegOut{block * pl.bins, seg_cap, seg_cap, seg_cap, seg_ovf};
        if (nsets > 1u && pl.out_lc) { sg.lc = pl.out_lc; sg.cblock = (uint32_t) block; sg.cbins = pl.bins; }
        cursor = leafcnt + block * pl.bins;
    }
    if (LEAF6) sg.n_total = (uint64_t) pl.nparts * pl.bins * seg_cap; // (level 2: one set of leaves per input partition)
    uint32_t run[2] = {0u, 0u};
    SegClk clk;
    if (SEGM) {
        for (uint32_t b = threadIdx.x; b <= pl.bins; b += blockDim.x) ls.cnt[b] = 0;
    } else {
        for (uint32_t b = threadIdx.x; b < pl.bins; b += blockDim.x) {
            l.gbase[b] = outbounds[(uint64_t) (blockIdx.x / pl.chunks) * pl.bins + b] + offs_rel[(uint64_t) blockIdx.x * pl.bins + b];
            l.lstart[b] = 0;
        }
        if (threadIdx.x == 0) l.lstart[pl.bins] = 0;
    }
    lds_barrier();
    uint64_t i0, i1;
    if (pl.seg_units) { // (positions in the partition's segments, one after the other)
        i0 = 0;
        i1 = (uint64_t) pl.seg_units * pl.seg_cap;
        if (SHARED) { // this unit's slice, from a multiple of 16 positions on
            const uint64_t per = ((i1 + pl.chunks - 1) / pl.chunks + 15) & ~(uint64_t) 15;
            i0 = (uint64_t) my_chunk * per < i1 ? (uint64_t) my_chunk * per : i1;
            i1 = i0 + per < i1 ? i0 + per : i1;
        }
    } else arr_unit_range(bounds, pl, blockIdx.x, &i0, &i1);
    // PADDED: the input is the segmented output of a single-pass level 1 (pl.seg_units != 0) -- a tile is requested whole by
    // unconditional loads (positions beyond the partition are mapped to its last block and not looked at) and the waits are
    // explicit (see flat_step_fetch); otherwise (arrays of the caller) the loads stay under their bounds tests and the
    // compiler's waits.
    constexpr bool PADDED = IT == IT_HASH && SEGM;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    uint64_t nxt[16];
    // PADDED: 16 bytes per lane and request -- item 2 j2 + e of a thread is element j2 * 2 THREADS + 2 tid + e of the tile (the
    // unit starts on a multiple of 16 items: segment sizes are multiples of 16)
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint64_t i = i0 + (uint64_t) j * blockDim.x + threadIdx.x;
        if (!PADDED) nxt[j] = i < i1 ? in[i] : CKEY_EMPTY;
    }
    // the 16 items of this thread of the tile that starts at position t of the partition
    const uint64_t seg_stride = (uint64_t) pl.seg_bins * pl.seg_cap, seg_base = (uint64_t) sp * pl.seg_cap;
    auto tile_request = [&](uint64_t t) {
        { // position -> (block, offset); a pair of items never straddles segments (their sizes are multiples of 16)
            uint32_t i = (uint32_t) t + 2u * threadIdx.x;
            uint32_t u = i / pl.seg_cap, o = i - u * pl.seg_cap;
#pragma unroll
            for (int j2 = 0; j2 < 8; j2++) {
                const uint32_t uu = u < pl.seg_units ? u : pl.seg_units - 1; // (beyond the partition: anything readable)
                const uint64_t at = pl.seg_lc ? seg_chunked_index(uu, pl.seg_bins, sp, pl.seg_cap, o, pl.seg_lc) : (uint64_t) uu * seg_stride + seg_base + o;
#if KMU_SCATTER_NTLOAD >= 1
                const u64x2 q = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in + at));
#else
                const u64x2 q = *reinterpret_cast<const u64x2 *>(in + at);
#endif
                nxt[2 * j2] = q.x;
                nxt[2 * j2 + 1] = q.y;
                o += 2u * THREADS;
                if (pl.seg_cap >= 2u * THREADS) { if (o >= pl.seg_cap) { o -= pl.seg_cap; u++; } }
                else { const uint32_t d = o / pl.seg_cap; u += d; o -= d * pl.seg_cap; }
            }
        }
    };
    if (PADDED) {
        if (i0 < i1) tile_request(i0);
        else {
#pragma unroll
            for (int j = 0; j < 16; j++) nxt[j] = CKEY_EMPTY;
        }
        vm_wait_all();
    }
    clk.start();
    for (uint64_t t0 = i0; t0 < i1; t0 += TILE) {
        uint64_t it[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (PADDED) it[j] = t0 + (uint64_t) (j >> 1) * (2u * THREADS) + 2u * threadIdx.x + (j & 1) < i1 ? nxt[j] : CKEY_EMPTY;
            else it[j] = nxt[j];
        }
        // the next tile is requested before this one is sorted: its HBM latency hides under the LDS work
        if (PADDED) {
            tile_request(t0 + TILE);
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint64_t i = t0 + TILE + (uint64_t) j * blockDim.x + threadIdx.x;
                nxt[j] = i < i1 ? in[i] : CKEY_EMPTY;
            }
        }
        if (IT == IT_KEY_TO_HASH) { // from here on the k-mers travel as their table hash (no further evaluations)
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (it[j] != CKEY_EMPTY) it[j] = khash(it[j]);
        }
        if (SEGM) tile_scatter_seg<PADDED, SHARED, THREADS, LEAF6, 3>(it, ls, pl.bins, pl.region_bits, pl.shift, out, sg, run, clk, cursor);
        else tile_scatter<IT == IT_KEY_TO_HASH ? IT_HASH : IT>(it, l, pl.bins, pl.region_bits, pl.shift, pl.bins - 1, out);
    }
    if (SEGM) {
        // a level with one unit per input partition writes leaves: their fills go to leafcnt, no tail marks
        if (!SHARED) seg_finish_unit(ls, pl.bins, sg, run, out, leafcnt == nullptr, leafcnt);
        clk.mark(5);
        clk.flush(1);
    }
}

// Level 1 of the receiver of a super-k-mer exchange (kmu_smer.h): the input is an array of 12-byte records, a thread takes one
// record per tile and expands it into its <= 16 canonical k-mers with the window arithmetic of the read path (a record IS the
// lane's three code words); from there on the tile sort of the single-pass partition, shared streams and cursors as in
// k_arr_scatter<IT_KEY_TO_HASH, true, SHARED>.  Unit u of `chunks` takes records [n u / chunks, n (u + 1) / chunks).
template <bool SHARED>
__global__ void __launch_bounds__(SCATTER_THREADS) k_smer_scatter1(const uint32_t *recs, uint64_t n_rec, int k, ArrPlan pl, uint64_t *out,
                                                                  uint64_t seg_cap, uint32_t *seg_ovf, uint32_t *cursors) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SegLds ls = seg_lds(smem, pl.bins);
    SegOut sg{(uint64_t) blockIdx.x * pl.bins, seg_cap, seg_cap, seg_cap, seg_ovf};
    uint32_t *cursor = nullptr;
    if (SHARED) {
        const uint32_t nsets = pl.out_sets > 1u ? pl.out_sets : 1u;
        const uint64_t block = nsets > 1u ? blockIdx.x % nsets : 0u;
        sg = SegOut{block * pl.bins, seg_cap, seg_cap, seg_cap, seg_ovf};
        if (nsets > 1u && pl.out_lc) { sg.lc = pl.out_lc; sg.cblock = (uint32_t) block; sg.cbins = pl.bins; }
        cursor = cursors + block * pl.bins;
    }
    uint32_t run[2] = {0u, 0u};
    SegClk clk;
    for (uint32_t b = threadIdx.x; b <= pl.bins; b += blockDim.x) ls.cnt[b] = 0;
    lds_barrier();
    const uint64_t i0 = n_rec * blockIdx.x / pl.chunks, i1 = n_rec * (blockIdx.x + 1) / pl.chunks;
    uint32_t nx0 = 0, nx1 = 0, nx2 = 0;
    bool nxv = false;
    auto fetch = [&](uint64_t i) {
        nxv = i < i1;
        if (nxv) { nx0 = recs[i * 3]; nx1 = recs[i * 3 + 1]; nx2 = recs[i * 3 + 2]; }
    };
    fetch(i0 + threadIdx.x);
    clk.start();
    for (uint64_t t0 = i0; t0 < i1; t0 += SCATTER_THREADS) {
        const uint32_t w0 = nx0, w1 = nx1, w2 = nx2, L = nxv ? (nx2 & 15u) + 1u : 0u;
        fetch(t0 + SCATTER_THREADS + threadIdx.x); // the next tile's record arrives under this tile's sort
        uint64_t it[16];
        const StepWin sw = step_win(w0, w1, w2 & ~15u, k); // (the low four bits of a record's last word: its k-mer count)
#pragma unroll
        for (int j = 0; j < 16; j++)
            it[j] = (uint32_t) j < L ? khash(step_canonical(sw, j)) : CKEY_EMPTY; // kmer.reverse_complement().min(kmer), kmercount.rs:938
        tile_scatter_seg<false, SHARED>(it, ls, pl.bins, pl.region_bits, pl.shift, out, sg, run, clk, cursor);
    }
    if (!SHARED) seg_finish_unit(ls, pl.bins, sg, run, out, true, nullptr);
    clk.mark(5);
    clk.flush(1);
}

// the overflow word block of a single-pass partition (seg_spill): flag and count zero, capacity and address of the list
__global__ void __launch_bounds__(64) k_spill_header(uint32_t *ovf, uint32_t cap, uint64_t *list) {
    if (threadIdx.x < 16) ovf[threadIdx.x] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        ovf[2] = cap;
        *reinterpret_cast<uint64_t **>(ovf + 4) = list;
    }
}

// shared segments (SegPlan1::sets): "no k-mer" marks from the fill of every (set, bin) stream to its capacity
__global__ void __launch_bounds__(256) k_seg_tails(const uint32_t *cursor, uint32_t cap, uint64_t *out, uint32_t lc, uint32_t nbins) {
    const uint32_t n = cursor[blockIdx.x] < cap ? cursor[blockIdx.x] : cap;
    for (uint32_t i = n + threadIdx.x; i < cap; i += blockDim.x)
        out[lc ? seg_chunked_index(blockIdx.x / nbins, nbins, blockIdx.x % nbins, cap, i, lc) : (uint64_t) blockIdx.x * cap + i] = CKEY_EMPTY;
}

// the spill list of a single-pass partition (khash values) into the finished table, by direct insertion
__global__ void __launch_bounds__(256) k_count_add_spill(const uint64_t *items, const uint32_t *ovf, CountTable t, uint32_t *err) {
    const uint32_t n = ovf[1] < ovf[2] ? ovf[1] : ovf[2];
    bool full = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (!count_insert_h(t, t.w ? 0ull : khash_inv(items[i]), items[i], 1u)) full = true;
    if (full) atomicOr(err, DERR_TABLE_FULL);
}

// bounds[i] = i * stride (the input partitions of the single-pass level 2: level 1's fixed-size bins)
__global__ void __launch_bounds__(256) k_fill_linear(uint64_t *out, uint64_t n, uint64_t stride) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = i * stride;
}

// build: one workgroup per region.  The region (keys + counts) lives in LDS while its k-mers are inserted, then it
// leaves for HBM.  in_mode: 0 = the table holds nothing yet (no region is read), 1 = open-addressing image (the slab is
// read as it is), 2 = compact (the region's pairs are re-inserted).  out_compact: only the occupied entries are written --
// rcount[r] (key, count) pairs at the head of the slab, in slot order -- instead of the whole 12-byte-per-slot image: at a
// load factor of 0.47 that is 48.6 GB instead of 103 GB per build of the bench workload.  The statistics of the table
// (distinct, unique, occurrences) fall out of the same pass (stats[0..2], one atomic per workgroup).
// The first BUILD_PRE items of every thread are requested before the region is initialised, so their HBM latency hides
// under the LDS fill; three workgroups share a CU and overlap each other's phases.
static constexpr int BUILD_THREADS = 512;
static constexpr int BUILD_PRE = 6;

// The region build for the quotient slot format (see the top of the file): the region is R 8-byte words in LDS (32 KiB for
// 4 096 slots: four workgroups of 512 threads per CU; 64 KiB for 8 192 slots: two workgroups of 1 024), a first sighting is
// one ds_cmpst_rtn_b64, a repeat one more ds_add_u64 (guarded: the count field stops at q_limit), and the LDS image leaves as
// it is.  in_mode: 0 = the table holds nothing yet, 1 = the slab is read first.  `flags`: A/B runs (KMU_BUILD_ABLATE).
// LEAF6: the leaves hold 6 bytes per item in two planes (tile_scatter_seg): bits 47..0 of the hash are all this kernel uses
// (hw = h << w, the home slot = the 12 bits below the region index)
template <int IT, int THREADS, bool LEAF6 = false>
__global__ void __launch_bounds__(THREADS) k_part_build_q(const uint64_t *__restrict__ items, const uint64_t *__restrict__ leafstart,
                                                          uint32_t n_regions, CountTable t, int in_mode, int flags, uint32_t *err,
                                                          uint64_t leaf_stride, const uint32_t *__restrict__ leafcnt, int contig) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint64_t n_total6 = (uint64_t) n_regions * leaf_stride;
    auto item_at = [&](uint64_t i) -> uint64_t { return LEAF6 ? leaf6_load(items, n_total6, i) : items[i]; };
    constexpr int BUILD_THREADS = THREADS; // (shadows the wide kernel's constant inside this function)
    const int out_compact = flags;
    const uint32_t R = t.rmask + 1;
    uint64_t *lk = reinterpret_cast<uint64_t *>(smem);
    uint4 *lk4 = reinterpret_cast<uint4 *>(lk);
    const uint32_t tid = threadIdx.x;
    uint32_t full = 0;
    const uint32_t r_begin = contig ? (uint32_t) ((uint64_t) blockIdx.x * n_regions / gridDim.x) : blockIdx.x;
    const uint32_t r_end = contig ? (uint32_t) ((uint64_t) (blockIdx.x + 1) * n_regions / gridDim.x) : n_regions;
    const uint32_t r_step = contig ? 1u : gridDim.x;
    const int w = t.w;
    const uint64_t cmask = q_cmask(w), add_limit = q_limit(w) - (THREADS > 512 ? (uint64_t) Q_MARGIN : 0ull);
    for (uint32_t r = r_begin; r < r_end; r += r_step) {
        const uint64_t i0 = leaf_stride ? (uint64_t) r * leaf_stride : leafstart ? leafstart[r] : 0;
        const uint64_t i1 = leaf_stride ? i0 + (leafcnt ? (leafcnt[r] < leaf_stride ? (uint64_t) leafcnt[r] : leaf_stride) : leaf_stride) : leafstart ? leafstart[r + 1] : 0;
        uint64_t pre_it[BUILD_PRE];
#pragma unroll
        for (int q = 0; q < BUILD_PRE; q++) {
            const uint64_t i = i0 + (uint64_t) q * BUILD_THREADS + tid;
            pre_it[q] = i < i1 ? item_at(i) : CKEY_EMPTY;
        }
        uint4 *gk4 = reinterpret_cast<uint4 *>(t.keys + (uint64_t) r * R);
        if (in_mode == 1) {
            for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) lk4[s] = gk4[s];
        } else {
            for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) lk4[s] = make_uint4(~0u, ~0u, ~0u, ~0u);
        }
        lds_barrier();
        // one probe of `item` at slot `off`: true = the item is in (claimed a free slot, or met its own key)
        auto probe = [&](uint64_t hw, uint32_t off) -> bool {
            const unsigned long long old = atomicCAS((unsigned long long *) &lk[off], (unsigned long long) CKEY_EMPTY, (unsigned long long) (hw | 1ull));
            if (old == CKEY_EMPTY) return true;
            if (!q_same(old, hw, w)) return false;
            // fewer than BUILD_THREADS adds are in flight behind a count seen below the limit: no carry into the key bits
            if ((old & cmask) < add_limit) atomicAdd((unsigned long long *) &lk[off], 1ull);
            return true;
        };
        if (!(out_compact & 32)) {
            // Every LANE walks through its prefetched items at its own pace: a lane whose item is in takes its next one in
            // the next trip of the loop.  (Item by item, a wave repeats the probe loop until the unluckiest of its 64 lanes
            // is through -- ~8 trips per item at a load factor of 0.47, ~30 per region -- while the lanes' SUMS of probes
            // over their four items lie close together.  Thread-0 clocks of the item-by-item form, r03: 79 % of a region's
            // time in this phase; KMU_BUILD_ABLATE=32 keeps that form for the A/B.)
            static_assert(BUILD_PRE == 6, "the item queue of a lane is written out by hand");
            // (two queues per lane with both probes in flight: 30.5 ms against 23.2 -- the loop is bound by the instructions
            //  of a trip, not by the LDS round trip)
            uint64_t q0 = pre_it[0], q1 = pre_it[1], q2 = pre_it[2], q3 = pre_it[3], q4 = pre_it[4], q5 = pre_it[5];
            uint32_t left = BUILD_PRE + 1, guard = 0, off = 0;
            uint64_t hw = 0;
            bool have = false; // this lane is probing for an item
            while (left) {
                if (!have) { // the lane's next item, if any ("no k-mer" marks are skipped a trip at a time)
                    const uint64_t item = q0;
                    q0 = q1; q1 = q2; q2 = q3; q3 = q4; q4 = q5; q5 = CKEY_EMPTY;
                    left--;
                    if (left && item != CKEY_EMPTY) {
                        const uint64_t h = IT == IT_HASH ? item : khash(item);
                        hw = h << w;
                        off = (uint32_t) (h >> t.shift) & t.rmask;
                        guard = 0;
                        have = true;
                    }
                }
                if (have) {
                    if (probe(hw, off)) have = false;
                    else {
                        off = (off + 1) & t.rmask;
                        if (++guard >= R) { full = 1; have = false; }
                    }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < BUILD_PRE; q++)
                if (pre_it[q] != CKEY_EMPTY) {
                    const uint64_t h = IT == IT_HASH ? pre_it[q] : khash(pre_it[q]);
                    uint32_t off = (uint32_t) (h >> t.shift) & t.rmask, n = 0;
                    while (!probe(h << w, off)) {
                        off = (off + 1) & t.rmask;
                        if (++n >= R) { full = 1; break; }
                    }
                }
        }
        for (uint64_t i = i0 + (uint64_t) BUILD_PRE * BUILD_THREADS + tid; i < i1; i += BUILD_THREADS) { // (leaves beyond 3 072 items)
            const uint64_t item = item_at(i);
            if (item == CKEY_EMPTY) continue;
            const uint64_t h = IT == IT_HASH ? item : khash(item);
            uint32_t off = (uint32_t) (h >> t.shift) & t.rmask, n = 0;
            while (!probe(h << w, off)) {
                off = (off + 1) & t.rmask;
                if (++n >= R) { full = 1; break; }
            }
        }
        lds_barrier();
        for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) gk4[s] = lk4[s];
        lds_barrier();
    }
    if (full) atomicOr(err, DERR_TABLE_FULL);
    return;
}

template <int IT>
__global__ void __launch_bounds__(BUILD_THREADS) k_part_build(const uint64_t *__restrict__ items, const uint64_t *__restrict__ leafstart,
                                                              uint32_t n_regions, CountTable t, int in_mode, int out_compact,
                                                              uint32_t *rcount, unsigned long long *stats, uint32_t *err,
                                                              uint64_t leaf_stride, const uint32_t *__restrict__ leafcnt, int contig) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t R = t.rmask + 1; // >= 1024
    // the regions of this workgroup: every gridDim.x-th one, or (contig) one contiguous range -- a workgroup then walks
    // through its own stretch of the leaves and of the table image
    const uint32_t r_begin = contig ? (uint32_t) ((uint64_t) blockIdx.x * n_regions / gridDim.x) : blockIdx.x;
    const uint32_t r_end = contig ? (uint32_t) ((uint64_t) (blockIdx.x + 1) * n_regions / gridDim.x) : n_regions;
    const uint32_t r_step = contig ? 1u : gridDim.x;
    uint64_t *lk = reinterpret_cast<uint64_t *>(smem);
    uint32_t *lc = reinterpret_cast<uint32_t *>(lk + R);
    uint64_t *bm = reinterpret_cast<uint64_t *>(lc + R); // 64 words: occupancy of the region's slots
    uint32_t *pre = reinterpret_cast<uint32_t *>(bm + 64); // 65 words: entries before each bitmap word, total
    uint4 *lk4 = reinterpret_cast<uint4 *>(lk), *lc4 = reinterpret_cast<uint4 *>(lc);
    const uint32_t tid = threadIdx.x, lane = (uint32_t) lane_id(), wave = tid >> 6;
    uint32_t full = 0;
    uint64_t st_d = 0, st_u = 0, st_o = 0;
    for (uint32_t r = r_begin; r < r_end; r += r_step) {
        const uint64_t gbase = (uint64_t) r * R;
        // the items of region r: [leafstart[r], leafstart[r + 1]), or a fixed-size range that may hold "no k-mer" marks (the
        // leaves of the single-pass partition), or none (the expansion of a compact table)
        const uint64_t i0 = leaf_stride ? (uint64_t) r * leaf_stride : leafstart ? leafstart[r] : 0;
        const uint64_t i1 = leaf_stride ? i0 + (leafcnt ? (leafcnt[r] < leaf_stride ? (uint64_t) leafcnt[r] : leaf_stride) : leaf_stride) : leafstart ? leafstart[r + 1] : 0;
        uint64_t pre_it[BUILD_PRE];
#pragma unroll
        for (int q = 0; q < BUILD_PRE; q++) {
            const uint64_t i = i0 + (uint64_t) q * BUILD_THREADS + tid;
            pre_it[q] = i < i1 ? items[i] : CKEY_EMPTY;
        }
        uint64_t *gk = t.keys + gbase;
        uint32_t *gc = t.counts + gbase;
        uint4 *gk4 = reinterpret_cast<uint4 *>(gk), *gc4 = reinterpret_cast<uint4 *>(gc);
        const uint32_t n_old = in_mode == 2 ? rcount[r] : 0u;
        if (in_mode == 1) {
            for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) lk4[s] = gk4[s];
            for (uint32_t s = tid; s < R / 4; s += BUILD_THREADS) lc4[s] = gc4[s];
        } else {
            for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) lk4[s] = make_uint4(~0u, ~0u, ~0u, ~0u);
            for (uint32_t s = tid; s < R / 4; s += BUILD_THREADS) lc4[s] = make_uint4(0u, 0u, 0u, 0u);
        }
        lds_barrier();
        auto insert_key = [&](uint64_t v, uint64_t h, uint32_t add) {
            uint32_t off = (uint32_t) (h >> t.shift) & t.rmask;
            bool done = false;
            for (uint32_t probes = 0; probes < R; probes++) {
                unsigned long long old = atomicCAS((unsigned long long *) &lk[off], (unsigned long long) CKEY_EMPTY,
                                                   (unsigned long long) v);
                if (old == CKEY_EMPTY || old == v) {
                    atomicAdd(&lc[off], add);
                    done = true;
                    break;
                }
                off = (off + 1) & t.rmask;
            }
            if (!done) full = 1;
        };
        auto insert = [&](uint64_t item) {
            if (IT == IT_HASH) insert_key(khash_inv(item), item, 1u);
            else insert_key(item, khash(item), 1u);
        };
        for (uint32_t i = tid; i < n_old; i += BUILD_THREADS) { // the pairs the region held (compact state)
            const uint64_t v = gk[i];
            insert_key(v, khash(v), gc[i]);
        }
#pragma unroll
        for (int q = 0; q < BUILD_PRE; q++)
            if (pre_it[q] != CKEY_EMPTY) insert(pre_it[q]);
        for (uint64_t i = i0 + (uint64_t) BUILD_PRE * BUILD_THREADS + tid; i < i1; i += BUILD_THREADS) {
            const uint64_t item = items[i];
            if (item != CKEY_EMPTY) insert(item);
        }
        lds_barrier();
        if (!out_compact) {
            for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) gk4[s] = lk4[s];
            for (uint32_t s = tid; s < R / 4; s += BUILD_THREADS) gc4[s] = lc4[s];
            lds_barrier();
            continue;
        }
        // ---- compact write-out: slot s = j * 512 + tid belongs to bitmap word j * 8 + wave ----
        const uint32_t nj = R / BUILD_THREADS, nwords = R / 64;
        uint64_t mybits[8];
        uint32_t mycnt[8];
        uint64_t mykey[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            mybits[j] = 0;
            mycnt[j] = 0;
            mykey[j] = CKEY_EMPTY;
            if ((uint32_t) j < nj) {
                const uint32_t s_ = (uint32_t) j * BUILD_THREADS + tid;
                mykey[j] = lk[s_];
                mycnt[j] = lc[s_];
                const bool occ = mykey[j] != CKEY_EMPTY && mycnt[j] != 0u; // (a zero count: an entry that left for its owner)
                mybits[j] = __ballot(occ);
                if (lane == 0) bm[(uint32_t) j * (BUILD_THREADS / 64) + wave] = mybits[j];
                if (!occ) mycnt[j] = 0u;
            }
        }
        lds_barrier();
        if (tid < 64) { // exclusive prefix of the words' populations
            const uint32_t c = tid < nwords ? (uint32_t) __popcll(bm[tid]) : 0u;
            const uint32_t incl = wave_incl_scan_u32(c);
            pre[tid] = incl - c;
            if (tid == 63) pre[64] = incl;
        }
        lds_barrier();
        const uint32_t total = pre[64];
        // the pairs move to the head of the LDS arrays (every thread holds its slots in registers by now) and leave with
        // 16-byte stores like the open image does
#pragma unroll
        for (int j = 0; j < 8; j++)
            if ((uint32_t) j < nj && mycnt[j]) {
                const uint32_t pos = pre[(uint32_t) j * (BUILD_THREADS / 64) + wave] + (uint32_t) __popcll(mybits[j] & ((1ull << lane) - 1ull));
                lk[pos] = mykey[j];
                lc[pos] = mycnt[j];
                st_u += mycnt[j] == 1u;
                st_o += mycnt[j];
            }
        if (tid == 0) { rcount[r] = total; st_d += total; }
        lds_barrier();
        for (uint32_t s = tid; s < (total + 1) / 2; s += BUILD_THREADS) gk4[s] = lk4[s];
        for (uint32_t s = tid; s < (total + 3) / 4; s += BUILD_THREADS) gc4[s] = lc4[s];
        lds_barrier(); // bm / pre / lk / lc are rewritten by the next region
    }
    if (full) atomicOr(err, DERR_TABLE_FULL);
    if (out_compact) { // one atomic per workgroup and statistic
        __shared__ unsigned long long acc[3];
        if (tid == 0) { acc[0] = 0; acc[1] = 0; acc[2] = 0; }
        __syncthreads();
        for (int o = 32; o >= 1; o >>= 1) {
            st_u += ((uint64_t) (uint32_t) __shfl_xor((int) (st_u >> 32), o, 64) << 32) | (uint32_t) __shfl_xor((int) (uint32_t) st_u, o, 64);
            st_o += ((uint64_t) (uint32_t) __shfl_xor((int) (st_o >> 32), o, 64) << 32) | (uint32_t) __shfl_xor((int) (uint32_t) st_o, o, 64);
        }
        if (lane == 0) { atomicAdd(&acc[1], (unsigned long long) st_u); atomicAdd(&acc[2], (unsigned long long) st_o); }
        if (tid == 0) acc[0] = st_d;
        __syncthreads();
        if (tid < 3 && acc[tid]) atomicAdd(&stats[tid], acc[tid]);
    }
}

} // namespace kmu

using namespace kmu;

// bytes of the table image in HBM (what a pass over the table reads)
// kmer_owner's mode for the counter's own owner function (distributed counters)
static int owner_mode_of(const kmu_counter *c) { return c->okind == 1 ? owner_mode_smer(c->p.kmer_size) : (kmer_val_bytes(c->p.kmer_type) == 4 ? 1 : 0); }
static size_t table_image_bytes(const kmu_counter *c) { return (size_t) c->nslots * (c->qw ? 8 : 12); }

static CountTable table_of(const kmu_counter *c) {
    CountTable t;
    t.keys = c->keys;
    t.counts = c->counts;
    t.shift = 64 - c->lg;
    t.rmask = (1u << c->rbits) - 1u;
    t.w = c->qw;
    t.rbits = c->rbits;
    return t;
}
static uint32_t max_count(const kmu_counter *c) { return c->p.counter_bits == 8 ? 255u : 65535u; }
static int grid_for(const kmu_ctx *ctx, uint64_t n, int per_block) {
    uint64_t blocks = (n + per_block - 1) / per_block;
    uint64_t cap = (uint64_t) ctx->num_cus * 8;
    return (int) std::max<uint64_t>(1, std::min(blocks, cap));
}

// The region build (and the expansion of a compact table: a build without items that writes the open image).
// `to_compact`: leave the table compact (the partitioned builds) or as the open-addressing image.
static size_t build_lds(const kmu_counter *c) { return c->qw ? (size_t) 8 << c->rbits : ((size_t) 12 << c->rbits) + 64 * 8 + 65 * 4 + 16; }
template <int IT>
static int launch_build(kmu_counter *c, const uint64_t *items, const uint64_t *leaves, bool to_compact, uint32_t *d_err,
                        uint64_t leaf_stride = 0, const uint32_t *leafcnt = nullptr, uint64_t n_items_hint = 0, bool leaf6 = false) {
    kmu_ctx *ctx = c->ctx;
    const uint64_t n_regions = c->nslots >> c->rbits;
    const int in_mode = c->empty ? 0 : c->compact ? 2 : 1;
    if (c->qw) to_compact = false; // (the compact state is a wide-format state)
    const int per_cu = !c->qw ? 3 : c->rbits == 13 ? 2 : 4;
    int grid = (int) std::min<uint64_t>(n_regions, (uint64_t) ctx->num_cus * per_cu * 8);
    int contig = 0; // A/B: KMU_BUILD_MAP=1: one contiguous range of regions per workgroup, as many workgroups as fit the chip; =2: eight times as many
    if (const char *e = getenv("KMU_BUILD_MAP")) {
        contig = atoi(e);
        if (contig == 1) grid = (int) std::min<uint64_t>(n_regions, (uint64_t) ctx->num_cus * per_cu);
    }
    if (to_compact) KMU_HIP(ctx, hipMemsetAsync(c->scalars + 4, 0, 24, ctx->stream));
    {
        KernelTimer tm(ctx, !items ? "k_part_expand" : c->qw ? "k_part_build_q" : "k_part_build"); // (the kernels' own names)
        size_t pad = 0; // A/B runs: extra dynamic LDS per workgroup (fewer workgroups per CU)
        if (const char *e = getenv("KMU_BUILD_LDS_PAD")) pad = (size_t) std::max(0, atoi(e));
        int abl = 0; // A/B: 32 = the prefetched items of a thread one after the other (the lanes of a wave in lock step)
        if (const char *e = getenv("KMU_BUILD_ABLATE")) abl = atoi(e);
        if (c->qw && c->rbits == 13) {
            const auto kq = k_part_build_q<IT, 1024>;
            if (!(ctx->lds_attr_set & (IT == IT_HASH ? 4u : 8u))) {
                KMU_HIP(ctx, hipFuncSetAttribute((const void *) kq, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                ctx->lds_attr_set |= IT == IT_HASH ? 4u : 8u;
            }
            hipLaunchKernelGGL(kq, dim3(grid), dim3(1024), build_lds(c) + pad, ctx->stream, items, leaves, (uint32_t) n_regions, table_of(c), in_mode, abl,
                               d_err, leaf_stride, leafcnt, contig);
        } else if (c->qw && leaf6)
            hipLaunchKernelGGL((k_part_build_q<IT, 512, true>), dim3(grid), dim3(512), build_lds(c) + pad, ctx->stream, items, leaves, (uint32_t) n_regions,
                               table_of(c), in_mode, abl, d_err, leaf_stride, leafcnt, contig);
        else if (c->qw)
            hipLaunchKernelGGL((k_part_build_q<IT, 512>), dim3(grid), dim3(512), build_lds(c) + pad, ctx->stream, items, leaves, (uint32_t) n_regions,
                               table_of(c), in_mode, abl, d_err, leaf_stride, leafcnt, contig);
        else
            hipLaunchKernelGGL((k_part_build<IT>), dim3(grid), dim3(BUILD_THREADS), build_lds(c), ctx->stream, items, leaves, (uint32_t) n_regions,
                               table_of(c), in_mode, to_compact ? 1 : 0, c->rcount, (unsigned long long *) (c->scalars + 4), d_err, leaf_stride, leafcnt, contig);
    }
    KMU_HIP(ctx, hipGetLastError());
    c->empty = false;
    c->compact = to_compact;
    c->stats_cached = to_compact;
    return KMU_OK;
}
// KMU_COUNT_COMPACT=1: the partitioned builds leave the table compact.  Measured on the bench workload (r02): 54 GB less
// written per build, and 30.1 ms instead of 26.5 -- the region build is bound by its phases (fill, insert, write-out, three
// workgroups per CU taking turns), not by the bytes it writes, and the compaction adds a phase.  The open image stays the
// default; the compact form is there for tables that are built once and then only asked for their statistics.
static bool want_compact(const kmu_counter *c) {
    if (c->qw) return false; // (a wide-format state)
    const char *e = getenv("KMU_COUNT_COMPACT");
    return e && atoi(e) != 0;
}

// make the table's open-addressing image physical before anything probes or updates slots in place: the logical "all
// free" state is written out, a compact table is expanded (one pass of the build kernel without items)
static int table_alloc(kmu_counter *c, uint64_t distinct_hint);
static int materialize(kmu_counter *c) {
    if (c->deferred) KMU_TRY(table_alloc(c, c->p.capacity_hint)); // (a reader before the first add: the table the hint asks for, empty)
    kmu_ctx *ctx = c->ctx;
    if (c->empty) {
        KMU_HIP(ctx, hipMemsetAsync(c->keys, 0xFF, c->nslots * 8, ctx->stream));
        if (!c->qw) KMU_HIP(ctx, hipMemsetAsync(c->counts, 0, c->nslots * 4, ctx->stream));
        c->empty = false;
        c->compact = false;
        return KMU_OK;
    }
    if (c->compact) {
        uint32_t *d_err;
        KMU_TRY(get_err_word(ctx, &d_err));
        KMU_TRY(launch_build<IT_KEY>(c, nullptr, nullptr, false, d_err));
    }
    c->stats_cached = false; // whoever asked for the image may change it
    return KMU_OK;
}

// ---- the single-pass partition -------------------------------------------------------------------------------------------
// The exact route reads everything twice per level: a histogram pass gives every unit exact private output ranges, then the
// scatter.  The k-mers travel as khash(k-mer), so the digits are uniform: a unit's share of a bin is its item count / bins
// with a standard deviation of sqrt(that).  Here every (unit, bin) gets a FIXED segment of mean + 5 sigma + 32 items (6 %
// over the mean at level 1 of the bench size, 13 % per region leaf), the scatters run without their histogram passes (-7.6
// and -6.0 ms), the unused tail of every level-1 segment is filled with "no k-mer" marks that level 2 skips, the leaves
// carry their fill in a count of their own.  An item that finds its segment full goes to a spill list that is inserted
// into the finished table (seg_spill, k_count_add_spill: k-mers that arrive many at a time -- a genome at coverage -- vary
// a bin's fill more than independent ones); only a full spill list (k-mers that the hash cannot spread: a genome of one
// repeated k-mer) raises a flag that is read before the build touches the table: the call then takes the exact route from
// scratch.  Level 1 without a global histogram is also what lets kmu_sketch_count partition a chunk while the next one is
// uploaded.
// the LDS-staged scatter kernels ask for more than 64 KiB of dynamic LDS: one attribute call per instantiation and device
// (function attributes are per device: remembered per context, not per process)
static int scatter_attrs(kmu_ctx *ctx) {
    if (ctx->lds_attr_set & 1u) return KMU_OK;
    const void *fns[] = {(const void *) k_part_scatter1<false>, (const void *) k_part_scatter1<true>, (const void *) k_part_scatter1<true, true>,
                         (const void *) k_arr_scatter<IT_HASH, false>, (const void *) k_arr_scatter<IT_HASH, true>, (const void *) k_arr_scatter<IT_HASH, true, true>, (const void *) k_arr_scatter<IT_HASH, true, true, 512>,
                         (const void *) k_arr_scatter<IT_KEY, false>, (const void *) k_arr_scatter<IT_KEY_TO_HASH, false>,
                         (const void *) k_arr_scatter<IT_KEY_TO_HASH, true>, (const void *) k_arr_scatter<IT_KEY_TO_HASH, true, true>,
                         (const void *) k_smer_scatter1<false>, (const void *) k_smer_scatter1<true>,
                         (const void *) k_arr_scatter<IT_HASH, true, true, SCATTER_THREADS, true>};
    for (const void *f : fns) KMU_HIP(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ctx->lds_attr_set |= 1u;
    return KMU_OK;
}

// the partition plan of a table (levels and their fan-out); false: more than two levels would be needed
static bool part_plan_for(const kmu_counter *c, PartPlan *pl) {
    memset(pl, 0, sizeof *pl);
    pl->region_bits = c->lg - c->rbits;
    if (pl->region_bits <= 11) { pl->b1 = pl->region_bits; pl->b2 = 0; }
    else { pl->b1 = (pl->region_bits + 1) / 2; pl->b2 = pl->region_bits - pl->b1; }
    if (const char *e = getenv("KMU_COUNT_B1")) { // A/B: the split of the region bits between the two levels
        const int b1 = atoi(e);
        if (pl->b2 && b1 >= 1 && b1 <= 11 && pl->region_bits - b1 >= 1 && pl->region_bits - b1 <= 11) { pl->b1 = b1; pl->b2 = pl->region_bits - b1; }
    }
    return pl->b1 <= 11 && pl->b2 <= 11;
}

// the overflow word block of a single-pass partition (see seg_spill) and its spill list: room for 1/32 of the items
static int seg_spill_setup(kmu_ctx *ctx, uint64_t n_items, bool spill_ok, void **ovf_out) {
    void *ovf, *sp;
    const uint64_t cap = spill_ok ? std::min<uint64_t>(n_items / 32 + 4096, 0x7FFFFFFFull) : 0;
    KMU_TRY(dev_buf(ctx, "cnt.seg_ovf", 64, &ovf));
    KMU_TRY(dev_buf(ctx, "cnt.spill", (size_t) cap * 8 + 64, &sp));
    hipLaunchKernelGGL(k_spill_header, dim3(1), dim3(64), 0, ctx->stream, (uint32_t *) ovf, (uint32_t) cap, (uint64_t *) sp);
    KMU_HIP(ctx, hipGetLastError());
    *ovf_out = ovf;
    return KMU_OK;
}
// after the build: the spilled items, if any (h_ovf: the two words read back before the build)
static int seg_spill_add(kmu_counter *c, const void *ovf, const uint32_t *h_ovf, uint32_t *d_err) {
    kmu_ctx *ctx = c->ctx;
    if (!h_ovf[1]) return KMU_OK;
    void *sp;
    KMU_TRY(dev_buf(ctx, "cnt.spill", 64, &sp)); // (the list seg_spill_setup made: same buffer, never smaller)
    KernelTimer tm(ctx, "k_count_add_spill");
    hipLaunchKernelGGL(k_count_add_spill, dim3(grid_for(ctx, h_ovf[1], 256)), dim3(256), 0, ctx->stream, (const uint64_t *) sp, (const uint32_t *) ovf,
                       table_of(c), d_err);
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

static bool seg_partition_wanted(uint64_t total_bases) {
    const char *e = getenv("KMU_COUNT_SEG"); // 0: always the exact two-pass levels (A/B); 2: also for small batches (tests)
    if (e && atoi(e) == 0) return false;
    if (e && atoi(e) == 2) return true;
    if (total_bases >> 35) return false; // (positions inside a level-1 bin and the cursors of the shared segments are 32-bit numbers:
                                         //  a set of streams sees at most total / sets items, whatever their bins)
    // (round 2: from 2^27 bases on -- the margins of a segment per unit and bin are a small share of the mean only for big batches.
    //  With streams shared by a set's units they are sixteen times smaller, and the route wins wherever a table has two levels:
    //  5.9 / 8.8 / 17.6 / 35 / 70 Mbases: 0.22 / 0.27 / 0.40 / 0.66 / 1.14 ms against 0.65 / 0.70 / 0.83 / 1.14 / 1.68 for the exact levels,
    //  scripts/r03_segthr.sh.  A partitioned add is a batch of at least 1/64 (empty table) or 1/10 of the table's slots -- partitioned_batch_wanted --, so this is nearly every one of them.)
    return total_bases >= (1ull << 22);
}
static bool seg_layout_bin() { // A/B runs: level 1's output with the ranges of the bins side by side instead of the units' blocks
    const char *e = getenv("KMU_COUNT_SEG_LAYOUT");
    return e && e[0] == 'b';
}
// log2 of the chunk of the CHUNKED level-1 streams (SegOut), 0 = contiguous streams (the default); KMU_COUNT_SEG_CHUNK (A/B runs).
// Round 4, one box: level 1 12.3 (contiguous) / 12.6 (chunks of 2^10 or 2^12 items) / 12.7 ms (2^8), level 2 17.0 / 17.0 / 17.2 / 18.1:
// where the address translation is not the bound, the index arithmetic of the chunks is all that shows (scripts/r04_segchunk.sh).
// Kept behind the switch for the boxes on which level 1 runs 15 % slower in every launch (profiles/r04f_*), where it could not be tried.
static uint32_t seg_chunk_log(uint64_t cap) {
    uint32_t lc = 0;
    if (const char *e = getenv("KMU_COUNT_SEG_CHUNK")) lc = (uint32_t) std::max(0, std::min(16, atoi(e)));
    if (lc && lc < 4) lc = 4;             // (a pair of items, 16 bytes, never straddles chunks; capacities are multiples of 16)
    while (lc && (cap >> lc) < 8) lc--;   // small streams: a chunk row would be most of the stream
    return lc < 4 ? 0 : lc;
}
static uint64_t seg_cap_for(double mean) {
    double pct = 1.0;
    if (const char *e = getenv("KMU_COUNT_SEG_PCT")) pct = std::max(0.01, atof(e) / 100.0); // tests: force overflows
    // 5 sigma of independent k-mers: three segments in ten million overflow, by a few items that the spill list takes
    // (7 sigma + 64 before the spill list existed, when an overflow cost the whole attempt: 17.5 ms either way at level 1)
    double sig = 5.0, add = 32.0;
    if (const char *e = getenv("KMU_COUNT_SEG_SIGMA")) sig = atof(e); // A/B runs
    const double cap = (mean + sig * std::sqrt(mean) + add) * pct;
    return ((uint64_t) cap + 15) & ~(uint64_t) 15; // whole 128-byte lines
}
struct SegPlan {
    uint32_t units1, steps_per_unit;
    uint64_t cap1, bincap1, cap2, leafcap;
    uint32_t chunks2;
    uint32_t shared2; // level 2: the chunks2 units of a bin share its leaves (cursors in leafcnt)
    uint32_t sets; // level 1 with shared segments: one stream per (set, bin), cap1 items each (0: a segment per unit and bin)
    uint32_t lc1 = 0; // ... CHUNKED (SegOut): cap1 is a multiple of 2^lc1
};
// shared segments (tile_scatter_seg with cursors).  Level 1: sets of streams, two per XCD (bench workload, same box: 15.8-16.1 ms;
// one per XCD 18.5, four 15.9-16.7, eight 20.3, one for the whole chip 19.8-20.1, a unit's own segments 17.6-21.5 in two states).
static uint32_t seg_sets_wanted() {
    const char *se = getenv("KMU_COUNT_SEG_SHARED");
    return seg_layout_bin() ? 0u : se ? (uint32_t) std::max(0, atoi(se)) : 16u;
}
// Level 2: units per level-1 bin that share the bin's leaves (0: one unit, its own leaves: 23.5-24.2 ms; shared by 1 / 2 / 4 / 8 /
// 16 / 32 / 64 / 128 units: 21.6 / 22.3 / 20.3-21.4 / 18.5 / 16.9-17.4 / 17.7-17.9 / 17.9 / 20.0)
static uint32_t seg_l2_units_wanted() {
    const char *se = getenv("KMU_COUNT_L2_SHARED");
    return se ? (uint32_t) std::max(0, atoi(se)) : 16u;
}
static SegPlan seg_plan(const kmu_ctx *ctx, uint64_t total_bases, const PartPlan &pl, bool chunked) {
    SegPlan sp;
    const uint64_t nsteps = ((total_bases + 15) / 16 + 63) / 64;
    const uint32_t bins1 = 1u << pl.b1, bins2 = 1u << pl.b2;
    // units of level 1: one workgroup per CU.  Under the upload of kmu_sketch_count (chunked) the same units take a slice of every
    // arrival (seg_level1: rounds); round 2 launched FINISHED units instead -- up to four times as many, smaller ones, a CU's worth
    // per arrival, margins of 18 %: 4 x 8 ms at the bench size against 19 ms in one launch (KMU_COUNT_SEG_ROUNDS=0 keeps that form
    // for the A/B)
    const uint64_t by_size = total_bases / ((uint64_t) bins1 * 2048) + 1;
    uint64_t want = (uint64_t) ctx->num_cus;
    const char *re = getenv("KMU_COUNT_SEG_ROUNDS");
    if (chunked && re && atoi(re) == 0) want = std::min<uint64_t>(std::max<uint64_t>(by_size, want), want * 4);
    if (const char *e = getenv("KMU_COUNT_SEG_UNITS")) want = (uint64_t) std::max(1, atoi(e)); // A/B runs
    sp.units1 = (uint32_t) std::min<uint64_t>(nsteps, want);
    sp.steps_per_unit = (uint32_t) ((nsteps + sp.units1 - 1) / sp.units1);
    sp.units1 = (uint32_t) ((nsteps + sp.steps_per_unit - 1) / sp.steps_per_unit);
    sp.cap1 = seg_cap_for((double) sp.steps_per_unit * 1024.0 / bins1); // a segment per (unit, bin): margins of 5.9 % at the bench size
    sp.bincap1 = (uint64_t) sp.units1 * sp.cap1;
    sp.sets = 0;
    if (const uint32_t want_sets = seg_sets_wanted()) { // a stream per (set, bin), shared by the set's units: margins of 1.4 %
        sp.sets = std::min(want_sets, sp.units1);
        const uint64_t units_per_set = (sp.units1 + sp.sets - 1) / sp.sets;
        sp.cap1 = seg_cap_for((double) units_per_set * sp.steps_per_unit * 1024.0 / bins1);
        sp.lc1 = seg_layout_bin() ? 0u : seg_chunk_log(sp.cap1);
        if (sp.lc1) sp.cap1 = (sp.cap1 + (1ull << sp.lc1) - 1) >> sp.lc1 << sp.lc1;
        sp.bincap1 = (uint64_t) sp.sets * sp.cap1;
    }
    sp.chunks2 = 1; // a leaf has ONE capacity, whoever fills it: a unit per level-1 bin, or (shared2) several through the leaf's cursor
    sp.cap2 = seg_cap_for((double) total_bases / bins1 / bins2);
    sp.leafcap = sp.cap2;
    sp.shared2 = 0;
    if (const uint32_t want = seg_l2_units_wanted()) { sp.shared2 = 1; sp.chunks2 = want; }
    return sp;
}
// state of a single-pass partition between its level-1 launches (kmu_sketch_count runs them chunk by chunk under the upload)
struct SegRun {
    PartPlan pl;
    SegPlan sp;
    DevSeqs ds;
    uint64_t total_bases = 0;
    uint32_t units_done = 0;
    uint64_t steps_done = 0; // rounds: wave steps of the stream that have been through level 1
    bool rounds = false;     // the units persist across the launches (chunked calls)
    void *state = nullptr, *step_tab = nullptr;
    void *A = nullptr, *B = nullptr, *ovf = nullptr, *bnd = nullptr, *leafcnt = nullptr;
    uint32_t *d_err = nullptr;
};
// own_buffer: the level-1 output must survive other users of the shared scratch "cnt.partA" (kmu_sketch_count's chunked form:
// the sketch of the next chunk writes its (key, weight) lists there while this partition is still being filled)
static int seg_begin(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, const PartPlan &pl_in, uint32_t *d_err, SegRun *run,
                     bool own_buffer = false) {
    kmu_ctx *ctx = c->ctx;
    run->pl = pl_in;
    run->sp = seg_plan(ctx, total_bases, pl_in, own_buffer);
    run->ds = ds;
    run->total_bases = total_bases;
    run->units_done = 0;
    run->steps_done = 0;
    {
        const char *re = getenv("KMU_COUNT_SEG_ROUNDS");
        run->rounds = own_buffer && !(re && atoi(re) == 0) && !seg_layout_bin();
    }
    run->d_err = d_err;
    const uint32_t bins1 = 1u << pl_in.b1;
    const uint64_t n_regions = 1ull << pl_in.region_bits;
    run->pl.units1 = run->sp.units1;
    run->pl.steps_per_unit = run->sp.steps_per_unit;
    // (+ two tiles: level 2 requests whole tiles a tile ahead, k_arr_scatter's PADDED form: its last request of the last bin
    // ends less than two tiles behind the bin)
    KMU_TRY(dev_buf(ctx, own_buffer ? "cnt.segA" : "cnt.partA", (size_t) bins1 * run->sp.bincap1 * 8 + (size_t) 2 * TILE_ITEMS * 8 + 64, &run->A));
    KMU_TRY(dev_buf(ctx, "cnt.partB", (size_t) n_regions * run->sp.leafcap * 8 + 64, &run->B));
    KMU_TRY(dev_buf(ctx, "cnt.leafcnt", (size_t) n_regions * 4 + 64, &run->leafcnt));
    KMU_TRY(seg_spill_setup(ctx, total_bases, !want_compact(c), &run->ovf));
    KMU_TRY(dev_buf(ctx, "cnt.seg_bounds", ((size_t) bins1 + 1) * 8, &run->bnd));
    if (run->sp.sets) {
        KMU_TRY(dev_buf(ctx, "cnt.seg_state", (size_t) run->sp.sets * bins1 * 4 + 64, &run->state));
        KMU_HIP(ctx, hipMemsetAsync(run->state, 0, (size_t) run->sp.sets * bins1 * 4, ctx->stream));
    } else if (run->rounds) KMU_TRY(dev_buf(ctx, "cnt.seg_state", (size_t) run->sp.units1 * bins1 * 4 + 64, &run->state));
    {
        const uint64_t nsteps = std::max<uint64_t>(1, ((total_bases + 15) / 16 + 63) / 64);
#if KMU_STEP_TAB == 2
        KMU_TRY(flat_novalid(ctx, ds, total_bases, c->p.kmer_size, nsteps * 64 + 64, "cnt.novalid", &run->step_tab));
#else
        KMU_TRY(dev_buf(ctx, "cnt.step_tab", (size_t) nsteps * 16 + 64, &run->step_tab));
        hipLaunchKernelGGL(k_step_table, dim3((unsigned) ((nsteps + 255) / 256)), dim3(256), 0, ctx->stream, ds.offsets, ds.n_seq, nsteps,
                           (uint4 *) run->step_tab);
#endif
    }
    KMU_TRY(scatter_attrs(ctx));
    return KMU_OK;
}
// level 1 for the units whose wave steps (and their 32-base halo) lie inside the first `bases_ready` bases of the stream
static int seg_level1(kmu_counter *c, SegRun *run, uint64_t bases_ready) {
    kmu_ctx *ctx = c->ctx;
    const uint64_t steps_ready = bases_ready >= run->total_bases ? ((run->total_bases + 15) / 16 + 63) / 64
                                                                 : (bases_ready >= 32 ? (bases_ready - 32) / 1024 : 0);
    if (run->rounds) { // every unit takes its slice of the wave steps that have arrived since the last launch
        if (run->units_done == run->sp.units1) return KMU_OK; // the last launch has been made
        const bool last = bases_ready >= run->total_bases;
        const uint64_t n_new = steps_ready > run->steps_done ? steps_ready - run->steps_done : 0;
        // an arrival of less than a few tiles per wave waits for the next one; the last launch is made in any case (it marks
        // the tails of the segments, over an empty range if nothing is left)
        const char *mn = getenv("KMU_COUNT_SEG_ROUND_MIN");
        const uint64_t min_steps = (uint64_t) run->sp.units1 * (mn ? (uint64_t) atoi(mn) : 64u);
        if (!last && n_new < min_steps) return KMU_OK;
        const uint32_t bins1 = 1u << run->pl.b1;
        PartPlan pl = run->pl;
        pl.steps_per_unit = (uint32_t) ((n_new + run->sp.units1 - 1) / run->sp.units1);
        if (pl.steps_per_unit == 0) pl.steps_per_unit = 1;
        {
            const auto k1 = run->sp.sets ? k_part_scatter1<true, true> : k_part_scatter1<true, false>;
            KernelTimer tm(ctx, "k_part_scatter1");
            hipLaunchKernelGGL(k1, dim3(run->sp.units1), dim3(SCATTER_THREADS), seg_lds_bytes(bins1), ctx->stream, run->ds.bases,
                               run->ds.offsets, run->ds.n_seq, c->p.kmer_size, pl, (const uint64_t *) nullptr, (const uint64_t *) nullptr,
                               (uint64_t *) run->A,
                               SegPlan1{run->sp.cap1, 0, 0, run->steps_done, (uint32_t *) run->ovf, run->d_err, run->steps_done + n_new,
                                        (uint32_t *) run->state, (run->steps_done == 0 ? 1 : 0) | (getenv("KMU_COUNT_SEG_SETMAP") ? 2 : 0), last ? 1 : 0, run->sp.sets,
                                        (const uint4 *) run->step_tab, run->sp.lc1});
        }
        KMU_HIP(ctx, hipGetLastError());
        run->steps_done += n_new;
        if (last) {
            run->units_done = run->sp.units1;
            if (run->sp.sets)
                hipLaunchKernelGGL(k_seg_tails, dim3(run->sp.sets * bins1), dim3(256), 0, ctx->stream, (const uint32_t *) run->state,
                                   (uint32_t) run->sp.cap1, (uint64_t *) run->A, run->sp.lc1, bins1);
        }
        return KMU_OK;
    }
    uint32_t upto = bases_ready >= run->total_bases ? run->sp.units1
                                                    : (uint32_t) std::min<uint64_t>(steps_ready / run->sp.steps_per_unit, run->sp.units1);
    if (bases_ready < run->total_bases) { // partial launches in whole rounds of one workgroup per CU: a launch of fewer leaves CUs idle
        const uint32_t round = (uint32_t) ctx->num_cus;
        if (upto < run->units_done + round) return KMU_OK;
        upto = run->units_done + (upto - run->units_done) / round * round;
    }
    if (upto <= run->units_done) return KMU_OK;
    const uint32_t bins1 = 1u << run->pl.b1;
    {
        const auto k1 = run->sp.sets ? k_part_scatter1<true, true> : k_part_scatter1<true, false>;
        KernelTimer tm(ctx, "k_part_scatter1");
        hipLaunchKernelGGL(k1, dim3(upto - run->units_done), dim3(SCATTER_THREADS), seg_lds_bytes(bins1), ctx->stream,
                           run->ds.bases, run->ds.offsets, run->ds.n_seq, c->p.kmer_size, run->pl, (const uint64_t *) nullptr,
                           (const uint64_t *) nullptr, (uint64_t *) run->A,
                           SegPlan1{run->sp.cap1, seg_layout_bin() ? run->sp.bincap1 : 0, run->units_done, (uint64_t) run->units_done * run->sp.steps_per_unit,
                                    (uint32_t *) run->ovf, run->d_err, 0, (uint32_t *) run->state, 1 | (getenv("KMU_COUNT_SEG_SETMAP") ? 2 : 0), 1, run->sp.sets,
                                    (const uint4 *) run->step_tab, run->sp.lc1});
    }
    KMU_HIP(ctx, hipGetLastError());
    run->units_done = upto;
    if (run->sp.sets && upto == run->sp.units1)
        hipLaunchKernelGGL(k_seg_tails, dim3(run->sp.sets * bins1), dim3(256), 0, ctx->stream, (const uint32_t *) run->state,
                           (uint32_t) run->sp.cap1, (uint64_t *) run->A, run->sp.lc1, bins1);
    return KMU_OK;
}
// level 2 of a single-pass partition: A (level 1's segments: ap.seg_*) -> the leaves in B; shared: ap.chunks units per bin share
// the bin's leaves (cursors in leafcnt).  Shared leaves free the tile size from the length of a run: with <= 1 024 bins a
// workgroup is 512 threads on tiles of 8 192 items (76 KiB of LDS), TWO per CU, and the LDS phases of one run under the memory
// phases of the other (thread-0 clocks of the one-workgroup form: rank / scan / place 65 %, loads + stores 35 %, one after the
// other) -- measured: 19.4-19.6 ms against 16.4-17.8 for one workgroup of 1 024 threads (18.4 / 17.8 with 32 / 64 units per bin):
// the level runs at 4 TB/s either way, the memory system's rate for this mix; KMU_COUNT_L2_THREADS=512 keeps the A/B
// leaf items of 6 bytes (two planes) instead of 8: a table whose region index takes w >= 16 hash bits (>= 2^28 slots, quotient
// slots, regions of 4096), leaves shared through cursors (no tail marks to write), 1 024-thread workgroups.  KMU_COUNT_LEAF6=0/1.
static bool leaf6_wanted(const kmu_counter *c, bool shared2) {
    // (same box, alternating processes, scripts/r04_leaf6.sh: the headline's build 23.5 -> 20.2 ms, its level 2 17.4 -> 18.9 -- two
    //  stores per item --, the count unit 55.3 -> 53.1; config 4's shard 9.26 -> 8.84)
    bool on = true;
    if (const char *e = getenv("KMU_COUNT_LEAF6")) on = atoi(e) != 0;
    const char *te = getenv("KMU_COUNT_L2_THREADS");
    return on && shared2 && c->qw >= 16 && c->rbits == 12 && !(te && atoi(te) == 512);
}
static int seg_launch_level2(kmu_ctx *ctx, const ArrPlan &ap, bool shared, const void *A, const void *bnd, void *B, uint64_t cap2, void *ovf,
                             void *leafcnt, bool leaf6 = false) {
    const uint32_t bins2 = ap.bins;
    if (shared) KMU_HIP(ctx, hipMemsetAsync(leafcnt, 0, (size_t) ap.nparts * bins2 * 4, ctx->stream));
    int threads = SCATTER_THREADS;
    if (shared && bins2 <= 1024u) {
        const char *te = getenv("KMU_COUNT_L2_THREADS");
        threads = te ? atoi(te) : 1024;
        if (threads != 512) threads = SCATTER_THREADS;
    }
    auto k2 = shared ? k_arr_scatter<IT_HASH, true, true> : k_arr_scatter<IT_HASH, true, false>;
    if (threads == 512) k2 = k_arr_scatter<IT_HASH, true, true, 512>;
    if (leaf6) k2 = k_arr_scatter<IT_HASH, true, true, SCATTER_THREADS, true>;
    KernelTimer tm(ctx, "k_arr_scatter");
    hipLaunchKernelGGL(k2, dim3(ap.nparts * ap.chunks), dim3(threads), seg_lds_bytes(bins2, 16u * (uint32_t) threads), ctx->stream, (const uint64_t *) A,
                       (const uint64_t *) bnd, ap, (const uint64_t *) nullptr, (const uint64_t *) nullptr, (uint64_t *) B, cap2, (uint32_t *) ovf,
                       (uint32_t *) leafcnt);
    return KMU_OK;
}
// the rest of level 1, level 2, the overflow flag, the build.  *taken = 0: a segment overflowed, the table is untouched.
static int seg_finish(kmu_counter *c, SegRun *run, int *taken) {
    kmu_ctx *ctx = c->ctx;
    *taken = 0;
    KMU_TRY(seg_level1(c, run, run->total_bases));
    const uint32_t bins1 = 1u << run->pl.b1, bins2 = 1u << run->pl.b2;
    hipLaunchKernelGGL(k_fill_linear, dim3(8), dim3(256), 0, ctx->stream, (uint64_t *) run->bnd, (uint64_t) bins1 + 1, run->sp.bincap1);
    {
        ArrPlan ap{run->pl.region_bits, 0, bins2, bins1, run->sp.chunks2, run->sp.sets ? run->sp.sets : run->sp.units1, (uint32_t) run->sp.cap1, bins1};
        if (seg_layout_bin()) { ap.seg_units = 1; ap.seg_cap = (uint32_t) run->sp.bincap1; }
        if (run->sp.sets) ap.seg_lc = run->sp.lc1;
        KMU_TRY(seg_launch_level2(ctx, ap, run->sp.shared2 != 0, run->A, run->bnd, run->B, run->sp.cap2, run->ovf, run->leafcnt,
                                  leaf6_wanted(c, run->sp.shared2 != 0 && !seg_layout_bin())));
    }
    KMU_HIP(ctx, hipGetLastError());
    uint32_t h_ovf[2] = {0, 0}; // read before the table is touched: a full spill list leaves the call to the exact route
    KMU_HIP(ctx, hipMemcpyAsync(h_ovf, run->ovf, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_ovf[0]) return KMU_OK;
    KMU_TRY(launch_build<IT_HASH>(c, (const uint64_t *) run->B, nullptr, want_compact(c), run->d_err, run->sp.leafcap, (const uint32_t *) run->leafcnt, 0,
                                  leaf6_wanted(c, run->sp.shared2 != 0 && !seg_layout_bin())));
    KMU_TRY(seg_spill_add(c, run->ovf, h_ovf, run->d_err));
#if KMU_DIAG
    if (getenv("KMU_DIAG_SEG")) { // thread-0 clocks of the tile sort, summed over the workgroups: [level] between / rank / scan / place / out / finish
        unsigned long long h[2][8], z[2][8] = {};
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        KMU_HIP(ctx, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_diag_seg), sizeof h));
        KMU_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_diag_seg), z, sizeof z));
        for (int lv = 0; lv < 2; lv++) {
            unsigned long long tot = 0;
            for (int i = 0; i < 6; i++) tot += h[lv][i];
            fprintf(stderr, "diag seg level %d:", lv + 1);
            for (int i = 0; i < 6; i++) fprintf(stderr, " %.1f%%", tot ? 100.0 * (double) h[lv][i] / (double) tot : 0.0);
            fprintf(stderr, "  (between rank scan place out finish; %.3g clocks)\n", (double) tot);
        }
    }
#endif
    *taken = 1;
    return KMU_OK;
}
static int seg_partitioned_add(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, const PartPlan &pl_in, uint32_t *d_err, int *taken) {
    SegRun run;
    KMU_TRY(seg_begin(c, ds, total_bases, pl_in, d_err, &run));
    return seg_finish(c, &run, taken);
}

// the radix-partitioned build over device-resident ASCII reads
static int partitioned_add(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t *d_err) {
    kmu_ctx *ctx = c->ctx;
    PartPlan pl;
    pl.owner_parts = 0;
    pl.owner_w32 = 0;
    if (!part_plan_for(c, &pl)) return fail(ctx, KMU_E_UNSUPPORTED, "table too large for the two-level partitioned build");
    bool dbg_split = false;
#if KMU_DIAG // timing experiments of diagnostic builds only (scripts/dbg_split.sh): incomplete partition, no build
    if (const char *e = getenv("KMU_DBG_SPLIT")) {
        int x = 0, y = 0;
        if (sscanf(e, "%d,%d", &x, &y) == 2 && x >= 1 && x <= 11 && y >= 1 && y <= 11 && x + y <= pl.region_bits) { pl.b1 = x; pl.b2 = y; pl.region_bits = x + y; dbg_split = true; }
    }
#endif
    const uint32_t bins1 = 1u << pl.b1, bins2 = 1u << pl.b2;
    const uint64_t nsteps = ((total_bases + 15) / 16 + 63) / 64;
    uint32_t units1 = (uint32_t) std::min<uint64_t>(nsteps, (uint64_t) ctx->num_cus * 8);
    if (units1 < 1) units1 = 1;
    pl.steps_per_unit = (uint32_t) ((nsteps + units1 - 1) / units1);
    units1 = (uint32_t) ((nsteps + pl.steps_per_unit - 1) / pl.steps_per_unit);
    pl.units1 = units1;
    pl.chunks2 = pl.b2 ? std::max<uint32_t>(1u, 16384u / bins1) : 1u;
#if KMU_DIAG
    if (const char *e = getenv("KMU_DBG_CHUNKS2")) pl.chunks2 = (uint32_t) std::max(1, atoi(e));
#endif
    const uint32_t units2 = bins1 * pl.chunks2;
    const uint64_t n_regions = 1ull << pl.region_bits;

    // the single-pass partition first, BEFORE this route takes its buffers: both use "cnt.partA" / "cnt.partB" with different
    // sizes, and a buffer that grows is freed and allocated anew (pointers taken earlier would dangle)
    if (pl.b2 && !dbg_split && !c->no_seg && seg_partition_wanted(total_bases)) {
        int taken = 0;
        KMU_TRY(seg_partitioned_add(c, ds, total_bases, pl, d_err, &taken));
        if (taken) return KMU_OK; // (else a segment overflowed -- very skewed k-mers -- and nothing was touched: the exact route)
    }
    void *A, *B = nullptr, *hist1, *offs1, *tot1, *binstart1, *hist2 = nullptr, *offs2 = nullptr, *leafstart = nullptr, *tot2 = nullptr;
    KMU_TRY(dev_buf(ctx, "cnt.partA", total_bases * 8 + 64, &A));
    KMU_TRY(dev_buf(ctx, "cnt.hist1", (size_t) units1 * bins1 * 4, &hist1));
    KMU_TRY(dev_buf(ctx, "cnt.offs1", (size_t) units1 * bins1 * 8, &offs1));
    KMU_TRY(dev_buf(ctx, "cnt.tot1", (size_t) bins1 * 8, &tot1));
    KMU_TRY(dev_buf(ctx, "cnt.binstart1", (size_t) (bins1 + 1) * 8, &binstart1));
    if (pl.b2) {
        KMU_TRY(dev_buf(ctx, "cnt.partB", total_bases * 8 + 64, &B));
        KMU_TRY(dev_buf(ctx, "cnt.hist2", (size_t) units2 * bins2 * 4, &hist2));
        KMU_TRY(dev_buf(ctx, "cnt.offs2", (size_t) units2 * bins2 * 8, &offs2));
        KMU_TRY(dev_buf(ctx, "cnt.leafstart", (size_t) (n_regions + 1) * 8, &leafstart));
        KMU_TRY(dev_buf(ctx, "cnt.tot2", (size_t) n_regions * 8, &tot2));
    }
    const int k = c->p.kmer_size;
    KMU_TRY(scatter_attrs(ctx));
    {
        KernelTimer tm(ctx, "k_part_hist1");
        hipLaunchKernelGGL(k_part_hist1, dim3(units1), dim3(256), bins1 * 4, ctx->stream, ds.bases, ds.offsets, ds.n_seq, k,
                           pl, (uint32_t *) hist1, d_err, SampleArgs{nullptr, nullptr, 0u, 0u});
    }
    {
        KernelTimer tm(ctx, "k_part_scan1");
        hipLaunchKernelGGL(k_part_scan1a, dim3(bins1), dim3(256), 0, ctx->stream, (const uint32_t *) hist1, pl,
                           (uint64_t *) offs1, (uint64_t *) tot1);
        hipLaunchKernelGGL(k_part_scan1b, dim3(1), dim3(256), 0, ctx->stream, (const uint64_t *) tot1, pl,
                           (uint64_t *) binstart1);
    }
    {
        KernelTimer tm(ctx, "k_part_scatter1");
        hipLaunchKernelGGL(k_part_scatter1<false>, dim3(units1), dim3(SCATTER_THREADS), scatter_lds_bytes(bins1), ctx->stream,
                           ds.bases, ds.offsets, ds.n_seq, k, pl, (const uint64_t *) offs1, (const uint64_t *) binstart1,
                           (uint64_t *) A, SegPlan1{0, 0, 0, 0, nullptr, nullptr, 0, nullptr, 1, 1, 0, nullptr});
    }
    const uint64_t *items = (const uint64_t *) A;
    const uint64_t *leaves = (const uint64_t *) binstart1;
    if (pl.b2) {
        ArrPlan ap{pl.region_bits, 0, bins2, bins1, pl.chunks2};
        {
            KernelTimer tm(ctx, "k_arr_hist");
            hipLaunchKernelGGL(k_arr_hist<IT_HASH>, dim3(units2), dim3(256), bins2 * 4, ctx->stream, (const uint64_t *) A,
                               (const uint64_t *) binstart1, ap, (uint32_t *) hist2);
        }
        {
            KernelTimer tm(ctx, "k_arr_scan");
            const uint32_t T = scan_threads_per_bin(ap.chunks), per_wg = 256u / T;
            hipLaunchKernelGGL(k_arr_scan_a, dim3(bins1 * ((bins2 + per_wg - 1) / per_wg)), dim3(256), 0, ctx->stream,
                               (const uint32_t *) hist2, ap, T, (uint64_t *) offs2, (uint64_t *) tot2);
            hipLaunchKernelGGL(k_arr_scan_b, dim3(bins1), dim3(256), 0, ctx->stream, (const uint64_t *) tot2,
                               (const uint64_t *) binstart1, ap, (uint64_t *) leafstart);
        }
        {
            KernelTimer tm(ctx, "k_arr_scatter");
            hipLaunchKernelGGL((k_arr_scatter<IT_HASH, false>), dim3(units2), dim3(SCATTER_THREADS), scatter_lds_bytes(bins2), ctx->stream,
                               (const uint64_t *) A, (const uint64_t *) binstart1, ap, (const uint64_t *) offs2,
                               (const uint64_t *) leafstart,
                               (uint64_t *) B, 0ull, (uint32_t *) nullptr, (uint32_t *) nullptr);
        }
        items = (const uint64_t *) B;
        leaves = (const uint64_t *) leafstart;
    }
    if (!dbg_split) KMU_TRY(launch_build<IT_HASH>(c, items, leaves, want_compact(c), d_err, 0, nullptr, total_bases));
    return KMU_OK;
}

namespace kmu {

// Partition a device array of u64 keys by the top `region_bits` bits of khash(key) into 2^region_bits leaves
// (<= 22 bits: two 11-bit passes).  Returns the partitioned copy and the leaf bounds (both in context scratch
// buffers, valid until the next partition call).
// hashed_out: the output items are khash(key) instead of the keys (what k_part_build<IT_HASH> takes).
int partition_u64(kmu_ctx *ctx, const uint64_t *in, uint64_t n, int region_bits, const uint64_t **items_out,
                  const uint64_t **bounds_out, bool hashed_out) {
    if (region_bits > 22) return fail(ctx, KMU_E_UNSUPPORTED, "too many partitions (2^%d)", region_bits);
    KMU_TRY(scatter_attrs(ctx));
    const int b1 = region_bits <= 11 ? region_bits : (region_bits + 1) / 2;
    const int b2 = region_bits - b1;
    void *b0;
    KMU_TRY(dev_buf(ctx, "arr.bounds0", 16, &b0));
    uint64_t h0[2] = {0, n};
    KMU_HIP(ctx, hipMemcpyAsync(b0, h0, 16, hipMemcpyHostToDevice, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream)); // h0 lives on the stack
    const uint64_t *bounds = (const uint64_t *) b0;
    const uint64_t *items = in;
    uint32_t nparts = 1;
    bool first_done = false;
    for (int level = 0; level < 2; level++) {
        const int bits = level == 0 ? b1 : b2;
        if (bits == 0) continue;
        const int shift = level == 0 ? b2 : 0;
        const uint32_t bins = 1u << bits;
        uint32_t chunks = level == 0 ? (uint32_t) std::min<uint64_t>(std::max<uint64_t>(1, n / 65536), 16384)
                                     : std::max<uint32_t>(1u, 16384u / nparts);
        ArrPlan ap{region_bits, shift, bins, nparts, chunks};
        const uint32_t units = nparts * chunks;
        void *hist, *offs, *outb, *outbuf, *tot;
        KMU_TRY(dev_buf(ctx, level == 0 ? "arr.tot0" : "arr.tot1", (size_t) nparts * bins * 8, &tot));
        KMU_TRY(dev_buf(ctx, level == 0 ? "arr.hist0" : "arr.hist1", (size_t) units * bins * 4, &hist));
        KMU_TRY(dev_buf(ctx, level == 0 ? "arr.offs0" : "arr.offs1", (size_t) units * bins * 8, &offs));
        KMU_TRY(dev_buf(ctx, level == 0 ? "arr.bounds1" : "arr.bounds2", ((size_t) nparts * bins + 1) * 8, &outb));
        KMU_TRY(dev_buf(ctx, level == 0 ? "cnt.partA" : "cnt.partB", n * 8 + 64, &outbuf));
        {
            KernelTimer tm(ctx, "k_arr_hist");
            const bool in_hash = hashed_out && first_done; // the first executed level still reads keys
            if (in_hash) hipLaunchKernelGGL(k_arr_hist<IT_HASH>, dim3(units), dim3(256), bins * 4, ctx->stream, items, bounds, ap, (uint32_t *) hist);
            else hipLaunchKernelGGL(k_arr_hist<IT_KEY>, dim3(units), dim3(256), bins * 4, ctx->stream, items, bounds, ap, (uint32_t *) hist);
        }
        {
            KernelTimer tm(ctx, "k_arr_scan");
            const uint32_t T = scan_threads_per_bin(chunks), per_wg = 256u / T;
            hipLaunchKernelGGL(k_arr_scan_a, dim3(nparts * ((bins + per_wg - 1) / per_wg)), dim3(256), 0, ctx->stream,
                               (const uint32_t *) hist, ap, T, (uint64_t *) offs, (uint64_t *) tot);
            hipLaunchKernelGGL(k_arr_scan_b, dim3(nparts), dim3(256), 0, ctx->stream, (const uint64_t *) tot, bounds, ap,
                               (uint64_t *) outb);
        }
        {
            KernelTimer tm(ctx, "k_arr_scatter");
            const size_t slds = scatter_lds_bytes(bins);
            if (!hashed_out)
                hipLaunchKernelGGL((k_arr_scatter<IT_KEY, false>), dim3(units), dim3(SCATTER_THREADS), slds, ctx->stream, items, bounds, ap,
                                   (const uint64_t *) offs, (const uint64_t *) outb, (uint64_t *) outbuf, 0ull, (uint32_t *) nullptr, (uint32_t *) nullptr);
            else if (!first_done)
                hipLaunchKernelGGL((k_arr_scatter<IT_KEY_TO_HASH, false>), dim3(units), dim3(SCATTER_THREADS), slds, ctx->stream, items, bounds,
                                   ap, (const uint64_t *) offs, (const uint64_t *) outb, (uint64_t *) outbuf, 0ull, (uint32_t *) nullptr, (uint32_t *) nullptr);
            else
                hipLaunchKernelGGL((k_arr_scatter<IT_HASH, false>), dim3(units), dim3(SCATTER_THREADS), slds, ctx->stream, items, bounds, ap,
                                   (const uint64_t *) offs, (const uint64_t *) outb, (uint64_t *) outbuf, 0ull, (uint32_t *) nullptr, (uint32_t *) nullptr);
            first_done = true;
        }
        KMU_HIP(ctx, hipGetLastError());
        items = (const uint64_t *) outbuf;
        bounds = (const uint64_t *) outb;
        nparts *= bins;
    }
    *items_out = items;
    *bounds_out = bounds;
    return KMU_OK;
}

} // namespace kmu

// the single-pass partition for an ARRAY of canonical k-mers (what the owner of a key range receives in the OCCURRENCES
// route of a distributed add): level 1 cuts the array into chunks, every (chunk, bin) a fixed segment, no histograms
// recs != nullptr: the input is n_rec super-k-mer records holding n k-mers (kmu_smer.h) instead of an array of n k-mers
static int seg_partitioned_add_kmers(kmu_counter *c, const uint64_t *d_kmers, uint64_t n, const PartPlan &pl, uint32_t *d_err, int *taken,
                                     const void *recs = nullptr, uint64_t n_rec = 0) {
    kmu_ctx *ctx = c->ctx;
    *taken = 0;
    const uint32_t bins1 = 1u << pl.b1, bins2 = 1u << pl.b2;
    const uint64_t n_regions = 1ull << pl.region_bits;
    const uint32_t chunks1 = (uint32_t) ctx->num_cus; // one unit per CU: the biggest segments, the smallest margins
    // (shared segments as in the read path: sets of level-1 streams with cursors, the leaves of a bin shared by level 2's units)
    const uint32_t sets = std::min(seg_sets_wanted(), chunks1), units2 = seg_l2_units_wanted();
    const uint32_t pieces = sets ? sets : chunks1;
    uint64_t cap1 = seg_cap_for((double) n / pieces / bins1);
    const uint32_t lc1 = sets ? seg_chunk_log(cap1) : 0u; // (CHUNKED streams, SegOut)
    if (lc1) cap1 = (cap1 + (1ull << lc1) - 1) >> lc1 << lc1;
    const uint64_t bincap1 = (uint64_t) pieces * cap1;
    const uint64_t cap2 = seg_cap_for((double) n / bins1 / bins2);
    void *A, *B, *ovf, *bnd, *b0, *leafcnt, *cur1 = nullptr;
    KMU_TRY(dev_buf(ctx, "cnt.partA", (size_t) bins1 * bincap1 * 8 + (size_t) 2 * TILE_ITEMS * 8 + 64, &A));
    KMU_TRY(dev_buf(ctx, "cnt.partB", (size_t) n_regions * cap2 * 8 + 64, &B));
    KMU_TRY(dev_buf(ctx, "cnt.leafcnt", (size_t) n_regions * 4 + 64, &leafcnt));
    KMU_TRY(seg_spill_setup(ctx, n, !want_compact(c), &ovf));
    KMU_TRY(dev_buf(ctx, "cnt.seg_bounds", ((size_t) bins1 + 1) * 8, &bnd));
    KMU_TRY(dev_buf(ctx, "arr.bounds0", 16, &b0));
    if (sets) {
        KMU_TRY(dev_buf(ctx, "cnt.seg_state", (size_t) sets * bins1 * 4 + 64, &cur1));
        KMU_HIP(ctx, hipMemsetAsync(cur1, 0, (size_t) sets * bins1 * 4, ctx->stream));
    }
    hipLaunchKernelGGL(k_fill_linear, dim3(1), dim3(256), 0, ctx->stream, (uint64_t *) b0, (uint64_t) 2, n);
    KMU_TRY(scatter_attrs(ctx));
    if (recs) {
        ArrPlan ap{pl.region_bits, pl.b2, bins1, 1u, chunks1, 0u, 0u, 0u, sets};
        ap.out_lc = lc1;
        const auto k1 = sets ? k_smer_scatter1<true> : k_smer_scatter1<false>;
        KernelTimer tm(ctx, "k_smer_scatter1");
        hipLaunchKernelGGL(k1, dim3(chunks1), dim3(SCATTER_THREADS), seg_lds_bytes(bins1), ctx->stream, (const uint32_t *) recs, n_rec, c->p.kmer_size,
                           ap, (uint64_t *) A, cap1, (uint32_t *) ovf, (uint32_t *) cur1);
    } else {
        ArrPlan ap{pl.region_bits, pl.b2, bins1, 1u, chunks1, 0u, 0u, 0u, sets};
        ap.out_lc = lc1;
        const auto k1 = sets ? k_arr_scatter<IT_KEY_TO_HASH, true, true> : k_arr_scatter<IT_KEY_TO_HASH, true, false>;
        KernelTimer tm(ctx, "k_arr_scatter");
        hipLaunchKernelGGL(k1, dim3(chunks1), dim3(SCATTER_THREADS), seg_lds_bytes(bins1), ctx->stream, d_kmers, (const uint64_t *) b0, ap,
                           (const uint64_t *) nullptr, (const uint64_t *) nullptr, (uint64_t *) A, cap1, (uint32_t *) ovf, (uint32_t *) cur1);
    }
    if (sets)
        hipLaunchKernelGGL(k_seg_tails, dim3(sets * bins1), dim3(256), 0, ctx->stream, (const uint32_t *) cur1, (uint32_t) cap1, (uint64_t *) A, lc1, bins1);
    hipLaunchKernelGGL(k_fill_linear, dim3(8), dim3(256), 0, ctx->stream, (uint64_t *) bnd, (uint64_t) bins1 + 1, bincap1);
    {
        ArrPlan ap{pl.region_bits, 0, bins2, bins1, units2 ? units2 : 1u, pieces, (uint32_t) cap1, bins1, 0u};
        ap.seg_lc = lc1;
        KMU_TRY(seg_launch_level2(ctx, ap, units2 != 0, A, bnd, B, cap2, ovf, leafcnt, leaf6_wanted(c, units2 != 0)));
    }
    KMU_HIP(ctx, hipGetLastError());
    uint32_t h_ovf[2] = {0, 0};
    KMU_HIP(ctx, hipMemcpyAsync(h_ovf, ovf, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_ovf[0]) return KMU_OK; // the table is untouched: the exact levels take over
    KMU_TRY(launch_build<IT_HASH>(c, (const uint64_t *) B, nullptr, want_compact(c), d_err, cap2, (const uint32_t *) leafcnt, 0, leaf6_wanted(c, units2 != 0)));
    KMU_TRY(seg_spill_add(c, ovf, h_ovf, d_err));
    *taken = 1;
    return KMU_OK;
}

// big batches of explicit canonical k-mers (device arrays): partition by region, then the LDS build
static int partitioned_add_kmers(kmu_counter *c, const uint64_t *d_kmers, uint64_t n, uint32_t *d_err) {
    kmu_ctx *ctx = c->ctx;
    {
        PartPlan pl;
        if (!c->no_seg && part_plan_for(c, &pl) && pl.b2 && seg_partition_wanted(n)) {
            int taken = 0;
            KMU_TRY(seg_partitioned_add_kmers(c, d_kmers, n, pl, d_err, &taken));
            if (taken) return KMU_OK;
        }
    }
    const uint64_t *items, *bounds;
    const int region_bits = c->lg - c->rbits;
    // (a table of a single region is not partitioned at all: the items stay keys)
    const bool hashed = region_bits > 0;
    KMU_TRY(partition_u64(ctx, d_kmers, n, region_bits, &items, &bounds, hashed));
    if (hashed) return launch_build<IT_HASH>(c, items, bounds, want_compact(c), d_err, 0, nullptr, n);
    return launch_build<IT_KEY>(c, items, bounds, want_compact(c), d_err, 0, nullptr, n);
}

// canonical k-mers of the reads, grouped by owner rank (one level of the partition machinery with digit = owner).
// Two halves, so that a distributed add can look at the duplication sample between them: the census (per-unit histogram of
// the owners + the sample), then the scatter.
struct OwnerPlan {
    PartPlan pl;
    void *hist1, *offs1, *tot1, *binstart1;
};
static int owner_census(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t n_parts, uint32_t *d_err, OwnerPlan *op,
                        const SampleArgs &sa) {
    kmu_ctx *ctx = c->ctx;
    if (n_parts == 0 || n_parts > 2048) return fail(ctx, KMU_E_BAD_ARG, "n_parts must be in 1..2048");
    PartPlan &pl = op->pl;
    memset(&pl, 0, sizeof pl);
    pl.owner_parts = n_parts;
    pl.owner_w32 = kmer_val_bytes(c->p.kmer_type) == 4;
    const uint64_t nsteps = ((total_bases + 15) / 16 + 63) / 64;
    uint32_t units1 = (uint32_t) std::min<uint64_t>(std::max<uint64_t>(nsteps, 1), (uint64_t) ctx->num_cus * 8);
    pl.steps_per_unit = (uint32_t) ((std::max<uint64_t>(nsteps, 1) + units1 - 1) / units1);
    units1 = (uint32_t) ((std::max<uint64_t>(nsteps, 1) + pl.steps_per_unit - 1) / pl.steps_per_unit);
    pl.units1 = units1;
    KMU_TRY(dev_buf(ctx, "cnt.hist1", (size_t) units1 * n_parts * 4, &op->hist1));
    KMU_TRY(dev_buf(ctx, "cnt.offs1", (size_t) units1 * n_parts * 8, &op->offs1));
    KMU_TRY(dev_buf(ctx, "cnt.tot1", (size_t) n_parts * 8, &op->tot1));
    KMU_TRY(dev_buf(ctx, "cnt.binstart1", (size_t) (n_parts + 1) * 8, &op->binstart1));
    const int k = c->p.kmer_size;
    const size_t lds = ((size_t) n_parts + 2) * 4 + (sa.list ? 16 + (size_t) SAMPLE_LDS * 8 : 0);
    {
        KernelTimer tm(ctx, "k_part_hist1");
        hipLaunchKernelGGL(k_part_hist1, dim3(units1), dim3(256), lds, ctx->stream, ds.bases, ds.offsets, ds.n_seq, k, pl,
                           (uint32_t *) op->hist1, d_err, sa);
    }
    {
        KernelTimer tm(ctx, "k_part_scan1");
        hipLaunchKernelGGL(k_part_scan1a, dim3(n_parts), dim3(256), 0, ctx->stream, (const uint32_t *) op->hist1, pl,
                           (uint64_t *) op->offs1, (uint64_t *) op->tot1);
        hipLaunchKernelGGL(k_part_scan1b, dim3(1), dim3(256), 0, ctx->stream, (const uint64_t *) op->tot1, pl,
                           (uint64_t *) op->binstart1);
    }
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}
static int owner_scatter(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, const OwnerPlan &op, uint64_t **dev_out) {
    kmu_ctx *ctx = c->ctx;
    void *out;
    KMU_TRY(dev_buf(ctx, "cnt.partB", total_bases * 8 + 64, &out));
    KMU_TRY(scatter_attrs(ctx));
    {
        KernelTimer tm(ctx, "k_part_scatter1");
        hipLaunchKernelGGL(k_part_scatter1<false>, dim3(op.pl.units1), dim3(SCATTER_THREADS), scatter_lds_bytes(op.pl.owner_parts), ctx->stream,
                           ds.bases, ds.offsets, ds.n_seq, c->p.kmer_size, op.pl, (const uint64_t *) op.offs1,
                           (const uint64_t *) op.binstart1, (uint64_t *) out, SegPlan1{0, 0, 0, 0, nullptr, nullptr, 0, nullptr, 1, 1, 0, nullptr});
    }
    KMU_HIP(ctx, hipGetLastError());
    *dev_out = (uint64_t *) out;
    return KMU_OK;
}
static int extract_by_owner(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t n_parts, uint64_t **dev_out,
                            uint64_t *bounds_host, uint32_t *d_err) {
    kmu_ctx *ctx = c->ctx;
    OwnerPlan op;
    KMU_TRY(owner_census(c, ds, total_bases, n_parts, d_err, &op, SampleArgs{nullptr, nullptr, 0u, 0u}));
    KMU_TRY(owner_scatter(c, ds, total_bases, op, dev_out));
    KMU_HIP(ctx, hipMemcpyAsync(bounds_host, op.binstart1, (size_t) (n_parts + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KMU_OK;
}

// the table of a counter for `distinct_hint` distinct canonical k-mers: slots = the power of two >= 1.5 x the hint (load <= 2/3)
static int table_alloc(kmu_counter *c, uint64_t distinct_hint) {
    kmu_ctx *ctx = c->ctx;
    const kmu_count_params *p = &c->p;
    uint64_t want = std::max<uint64_t>(1024, distinct_hint + distinct_hint / 2);
    c->lg = 10;
    while ((1ull << c->lg) < want) c->lg++;
    // slot format: quotient (8 bytes per slot) whenever the region bits leave room for the count field (see the top of the
    // file): w = lg - 12 >= 11 for 8-bit counters, >= 17 for 16-bit ones.  KMU_COUNT_FMT=wide keeps 12 bytes per slot;
    // =quot (tests) raises a small table to the size the quotient format starts at.
    const int q_need = 12 + (p->counter_bits == 8 ? 11 : 17);
    bool quot = c->lg >= q_need;
    if (const char *e = getenv("KMU_COUNT_FMT")) {
        if (!strcmp(e, "wide")) quot = false;
        if (!strcmp(e, "quot")) { quot = true; c->lg = std::max(c->lg, q_need); }
    }
    c->nslots = 1ull << c->lg;
    c->rbits = std::min(c->lg, REGION_BITS_MAX);
    // quotient tables whose count field can spare a bit take regions of 8 192 slots (64 KiB of LDS, 1 024 threads): half as
    // many leaves, so that both partition levels of the bench's table fan out 1 024 ways (KMU_COUNT_RBITS=12 / 13: A/B)
    if (quot && c->lg - 13 >= q_need - 12 + 1) { // (one bit more: 1 024 threads may have 1 023 plain adds in flight, see k_part_build_q)
        const char *e = getenv("KMU_COUNT_RBITS");
        if (e && atoi(e) == 13) c->rbits = 13;
    }
    c->qw = quot ? c->lg - c->rbits : 0;
    hipError_t e1 = hipMalloc((void **) &c->keys, c->nslots * 8);
    hipError_t e2 = e1 == hipSuccess && !c->qw ? hipMalloc((void **) &c->counts, c->nslots * 4) : e1;
    if (e2 == hipSuccess) e2 = hipMalloc((void **) &c->rcount, ((c->nslots >> c->rbits) + 1) * 4);
    if (e2 != hipSuccess) {
        if (c->keys) (void) hipFree(c->keys);
        if (c->counts) (void) hipFree(c->counts);
        c->keys = nullptr;
        c->counts = nullptr;
        const unsigned long long ns = c->nslots;
        c->nslots = 0;
        (void) hipGetLastError();
        return fail(ctx, KMU_E_OOM, "cannot allocate a %llu-slot count table", ns);
    }
    c->empty = true;
    c->deferred = false;
    return KMU_OK;
}

// KMU_COUNT_HINT_OCCURRENCES: the table is allocated when the first k-mers come.  `ds` (unpacked reads, or null): their
// duplication is measured first -- a sample by key of the batch (k_part_hist1's: every occurrence of a sampled k-mer is in it),
// distinct keys of the sample counted in a scratch table -- and the table holds 1.5 x occurrences / ratio slots instead of
// 1.5 x occurrences: config 4's shard (0.75 G occurrences of 0.207 G distinct 31-mers) gets 2^29 slots = 4.3 GB where the
// occurrences ask for 2^31 = 17 GB, nine tenths of it empty, all of it written by every build.  One pass over the reads, once
// per counter.  `known_ratio` > 0: the caller has measured it (a distributed add); `occurrences`: the k-mers this table is to
// hold (0: the hint of kmu_count_create).
// occurrences / distinct of the k-mers of a batch of reads, from the key sample of a census pass (0: no estimate)
static int sample_ratio(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t *d_err, double *ratio_out) {
    kmu_ctx *ctx = c->ctx;
    *ratio_out = 0.0;
    void *slist, *sn, *stab;
    const uint64_t units = std::min<uint64_t>(std::max<uint64_t>(1, ((total_bases + 15) / 16 + 63) / 64), (uint64_t) ctx->num_cus * 8);
    const uint64_t kmers_per_unit = (total_bases + units - 1) / units;
    uint32_t shift = 0; // ~1024 sampled k-mers per workgroup (its LDS list holds 4096)
    while (shift < 24 && (kmers_per_unit >> shift) > 1024) shift++;
    const uint32_t cap = (uint32_t) std::min<uint64_t>(units * SAMPLE_LDS, 1u << 26);
    KMU_TRY(dev_buf(ctx, "cnt.sample", (size_t) cap * 8 + 64, &slist));
    KMU_TRY(dev_buf(ctx, "cnt.sample_n", 64, &sn));
    KMU_HIP(ctx, hipMemsetAsync(sn, 0, 64, ctx->stream));
    OwnerPlan op;
    KMU_TRY(owner_census(c, ds, total_bases, 1, d_err, &op, SampleArgs{(uint64_t *) slist, (uint32_t *) sn, cap, shift}));
    uint32_t h_sn[2] = {0, 0};
    KMU_HIP(ctx, hipMemcpyAsync(h_sn, sn, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t n_s = std::min(h_sn[0], cap);
    if (!n_s || h_sn[1]) return KMU_OK; // nothing sampled, or a truncated sample: no estimate
    uint32_t tbits = 10, d_s = 0;
    while ((1ull << tbits) < 2ull * n_s) tbits++;
    KMU_TRY(dev_buf(ctx, "cnt.sample_tab", ((size_t) 8 << tbits) + 64, &stab));
    KMU_HIP(ctx, hipMemsetAsync(stab, 0xFF, (size_t) 8 << tbits, ctx->stream));
    KMU_HIP(ctx, hipMemsetAsync((uint32_t *) sn + 2, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_sample_distinct, dim3(grid_for(ctx, n_s, 256)), dim3(256), 0, ctx->stream, (const uint64_t *) slist, n_s, (uint64_t *) stab,
                       (uint32_t) ((1u << tbits) - 1u), (uint32_t *) sn + 2);
    KMU_HIP(ctx, hipMemcpyAsync(&d_s, (uint32_t *) sn + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (d_s) *ratio_out = (double) n_s / (double) d_s;
    return KMU_OK;
}
static int table_alloc_for(kmu_counter *c, const DevSeqs *ds, uint64_t total_bases, uint32_t *d_err, double known_ratio = 0.0, uint64_t occurrences = 0) {
    if (!c->deferred) return KMU_OK;
    double ratio = known_ratio;
    if (ratio <= 0.0 && ds && !ds->packed && total_bases >= (1u << 16)) KMU_TRY(sample_ratio(c, *ds, total_bases, d_err, &ratio));
    if (const char *e = getenv("KMU_COUNT_DUP_RATIO")) ratio = atof(e); // (tests: the decision without the data)
    const uint64_t occ = occurrences ? occurrences : std::max<uint64_t>(c->p.capacity_hint, ds && !ds->packed ? total_bases : 0);
    uint64_t distinct = ratio > 1.0 ? (uint64_t) ((double) occ / ratio * 1.05) + 1024 : occ; // (5 %: the sample's error)
    return table_alloc(c, std::min(distinct, occ));
}

// Is a batch of n k-mers worth the streaming (partitioned) build?  It rewrites the table image, direct insertion costs ~52 ps per
// k-mer whatever the table: at the bench table (2^33 slots, scripts/r03_incr.py) a batch of 1/8 of the slots takes 37.8 ms partitioned
// against 67.6 direct, one of 1/16 33.0 against 28.3; into an EMPTY table (no image to read, and direct insertion has to wipe it
// first) 19.2 against 36.9 at 1/16 of the slots, 16.0 against 23.4 at 1/32, 13.9 against 18.0 at 1/64.  One rule for every way in
// (reads, k-mer arrays, super-k-mer records, the chunked host leg).
static bool partitioned_batch_wanted(const kmu_counter *c, uint64_t n) { return n * (c->empty ? 64u : 10u) >= c->nslots && n >= (1u << 16); }

static int add_entries(kmu_counter *c, const uint64_t *kmers, const uint32_t *counts, uint64_t n, int mem);
static int add_superkmers(kmu_counter *c, const void *recs, uint64_t n_rec, uint64_t n_kmers);
static int partitioned_add(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t *d_err);

// the canonical k-mers of device-resident reads into THIS table: the streaming build for big batches of unpacked reads,
// direct insertion otherwise (total_bases: extent of the flat stream, 0 for packed input)
static int local_add(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t *d_err) {
    kmu_ctx *ctx = c->ctx;
    KMU_TRY(table_alloc_for(c, &ds, total_bases, d_err));
    bool partitioned = false;
    const char *force = getenv("KMU_COUNT_PATH"); // "direct" / "partitioned": diagnostics
    if (!ds.packed) {
        partitioned = partitioned_batch_wanted(c, total_bases);
        if (force && !strcmp(force, "direct")) partitioned = false;
        if (force && !strcmp(force, "partitioned")) partitioned = total_bases > 0;
    }
    if (partitioned) return partitioned_add(c, ds, total_bases, d_err);
    KMU_TRY(materialize(c));
    CountTable t = table_of(c);
    if (!ds.packed) {
        int grid = ctx->num_cus * 8;
        KernelTimer tm(ctx, "k_count_add_flat");
        hipLaunchKernelGGL(k_count_add_flat, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets, ds.n_seq, c->p.kmer_size, t,
                           d_err);
    } else {
        int grid = (int) std::min<uint64_t>(ds.n_seq, (uint64_t) ctx->num_cus * 8);
        KernelTimer tm(ctx, "k_count_add_reads");
        hipLaunchKernelGGL(k_count_add_reads, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets, ds.packed_offsets, ds.n_seq,
                           ds.packed, ds.total_bytes, c->p.kmer_size, t, d_err);
    }
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}

// ---- distributed counting (KMU_COUNT_DISTRIBUTED; kmu.h "multi-GPU") ---------------------------------------------------------
// Cost model of the two routes for a rank's batch of n k-mer occurrences holding d distinct k-mers, in picoseconds per item,
// measured on one MI355X with the single-rank communicator (scripts/routes.sh; r03c: gpurun_out/routes_r03c.txt, ONT-shaped 4.36 G
// k-mers / config 4's 0.75 G): the owner census + duplication sample 2.4 / 4.1 per occurrence (both routes pay it), the scatter by
// owner 5.9 / 7.5, building a table from received keys 13.1 / 13.0 per key (two single-pass array-partition levels with shared
// segments + region build; 17.3 before round 3), the local build from reads 13.2 / 15.5 per occurrence (16.4 before), two passes
// over the table image for the export of MERGE (17.5 ms per pass over 68.7 GB: 4.0 TB/s), a received (k-mer, count) entry
// added by direct insertion ~60 / entry; the links: KMU_XGMI_GBPS per GPU and direction, all peers at once (default 350: seven
// links of 153 GB/s at ~1/3 efficiency until a multi-GPU measurement replaces it).
// Every input is either gathered from all ranks or a constant: the ranks MUST arrive at the same route (a rank on
// OCCURRENCES enters an all-to-all that a rank on MERGE does not).  table_bytes: the largest table image among the ranks;
// gbps: rank 0's KMU_XGMI_GBPS.
static double local_xgmi_gbps() {
    const char *e = getenv("KMU_XGMI_GBPS");
    return e && atof(e) > 0 ? atof(e) : 350.0;
}
static void route_model(double table_bytes, double gbps, double n, double d, int nranks, double *ms_occ, double *ms_merge) {
    const double f = nranks > 1 ? (double) (nranks - 1) / nranks : 0.0;
    const double ps = 1e-9; // ps -> ms
    // (the share a rank keeps travels too, through the device-to-device copy of the all-to-all's own slot: 35 GB in 29 ms at the
    //  bench size with one rank; MERGE's second table pass, the emit, only runs when something leaves)
    *ms_occ = n * (3.0 + 6.2 + 13.1) * ps + 8.0 * n * f / (gbps * 1e6) + 8.0 * n * (1.0 - f) / 1.2e9;
    *ms_merge = n * (3.0 + 13.5) * ps + (f > 0 ? 2.0 : 1.0) * table_bytes / 4.0e9 + 12.0 * d * f / (gbps * 1e6) + d * f * 60.0 * ps;
}
// The same with minimizer owners: the occurrences travel as `recs` super-k-mer records of 12 bytes (kmu_smer.h).  Per occurrence
// (one MI355X, one-rank communicator, profiles/r04b_routes.txt: the headline's 4.36 G k-mers / config 4's 0.75 G): census 1.5 / 2.0
// ps + record scatter 1.3 / 1.7 ps (neither forms a k-mer), the receiver's build from records 11.4 / 9.3 ps (record expansion inside
// the first partition level + level 2 + region build); MERGE pays the census, the local build (11.5), two table passes that
// compute a MINIMIZER per entry (24 ps per entry and pass: k_owner_census 99 ms over 4.05 G entries) and the direct merge.
static void route_model_smer(double table_bytes, double gbps, double n, double recs, double d, int nranks, double *ms_occ, double *ms_merge) {
    const double f = nranks > 1 ? (double) (nranks - 1) / nranks : 0.0;
    const double ps = 1e-9;
    *ms_occ = n * (1.5 + 1.3 + 11.4) * ps + 12.0 * recs * f / (gbps * 1e6) + 12.0 * recs * (1.0 - f) / 1.2e9;
    *ms_merge = n * (1.5 + 11.5) * ps + (f > 0 ? 2.0 : 1.0) * (table_bytes / 4.0e9 + d * 24.0 * ps) + 12.0 * d * f / (gbps * 1e6) + d * f * 60.0 * ps;
}

// first half of a distributed add: census of the owners + duplication sample, agreement on the route over all ranks, then
// OCCURRENCES / SUPERKMERS: scatter by owner and the all-to-all (on the communicator's stream; the context's stream is free for
// other work until dist_add_end), MERGE: the local build.
static int dist_add_begin(kmu_counter *c, DevSeqs &ds, uint64_t total_bases, uint32_t *d_err) {
    kmu_ctx *ctx = c->ctx;
    kmu_comm *cm = ctx->comm;
    if (!cm) return fail(ctx, KMU_E_BAD_ARG, "the communicator of this distributed counter's context is gone (kmu_comm_destroy)");
    const uint32_t N = (uint32_t) cm->nranks;
    const bool smer = c->okind == 1;
    comm_stats_reset(ctx);
    cm->stats.owner_kind = c->okind;
    // ---- census + sample ----
    void *slist, *sn, *stab;
    const uint64_t units = smer ? smer_units(ctx, total_bases)
                                : std::min<uint64_t>(std::max<uint64_t>(1, ((total_bases + 15) / 16 + 63) / 64), (uint64_t) ctx->num_cus * 8);
    const uint64_t kmers_per_unit = (total_bases + units - 1) / units;
    uint32_t shift = 0; // ~1024 sampled k-mers per workgroup (its LDS list holds 4096)
    while (shift < 24 && (kmers_per_unit >> shift) > 1024) shift++;
    const uint32_t cap = (uint32_t) std::min<uint64_t>(units * SAMPLE_LDS, 1u << 26);
    KMU_TRY(dev_buf(ctx, "cnt.sample", (size_t) cap * 8 + 64, &slist));
    KMU_TRY(dev_buf(ctx, "cnt.sample_n", 64, &sn));
    KMU_HIP(ctx, hipMemsetAsync(sn, 0, 64, ctx->stream));
    OwnerPlan op;
    SmerGroups sg;
    const SampleArgs sa{(uint64_t *) slist, (uint32_t *) sn, cap, shift};
    std::vector<uint64_t> bounds(N + 1), kto(N, 0); // groups (k-mers or records) per owner, as a prefix; k-mers per owner
    if (smer) {
        KMU_TRY(smer_census(ctx, ds, total_bases, c->p.kmer_size, N, d_err, sa, &sg));
        KMU_HIP(ctx, hipMemcpyAsync(bounds.data(), sg.binstart, (size_t) (N + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipMemcpyAsync(kto.data(), sg.kmers, (size_t) N * 8, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        KMU_TRY(owner_census(c, ds, total_bases, N, d_err, &op, sa));
        KMU_HIP(ctx, hipMemcpyAsync(bounds.data(), op.binstart1, (size_t) (N + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    uint32_t h_sn[4] = {0, 0, 0, 0};
    KMU_HIP(ctx, hipMemcpyAsync(h_sn, sn, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!smer)
        for (uint32_t p = 0; p < N; p++) kto[p] = bounds[p + 1] - bounds[p];
    uint64_t n_local = 0;
    for (uint32_t p = 0; p < N; p++) n_local += kto[p];
    const uint32_t n_s = std::min(h_sn[0], cap);
    uint32_t d_s = 0;
    if (n_s && !h_sn[1]) {
        uint32_t tbits = 10;
        while ((1ull << tbits) < 2ull * n_s) tbits++;
        KMU_TRY(dev_buf(ctx, "cnt.sample_tab", ((size_t) 8 << tbits) + 64, &stab));
        KMU_HIP(ctx, hipMemsetAsync(stab, 0xFF, (size_t) 8 << tbits, ctx->stream));
        KMU_HIP(ctx, hipMemsetAsync((uint32_t *) sn + 2, 0, 4, ctx->stream));
        {
            KernelTimer tm(ctx, "k_sample_distinct");
            hipLaunchKernelGGL(k_sample_distinct, dim3(grid_for(ctx, n_s, 256)), dim3(256), 0, ctx->stream, (const uint64_t *) slist, n_s,
                               (uint64_t *) stab, (uint32_t) ((1u << tbits) - 1u), (uint32_t *) sn + 2);
        }
        KMU_HIP(ctx, hipMemcpyAsync(&d_s, (uint32_t *) sn + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // ---- the ranks agree: sampled occurrences / distinct, local k-mers, and the send counts of every rank ----
    // (a k-mer's occurrences on different ranks are in different samples: the global distinct count of the sample is not
    //  known; the per-rank ratio is what decides how much MERGE saves on each rank, and the sum of both sides is used)
    // Whatever differs between the ranks and enters the decision travels in the row: the table size (capacity hints from
    // rank-specific read counts can straddle a power of two), the link rate and a forced route (environment of the rank's
    // process).  The ranks then evaluate the same function of the same gathered numbers: the largest table, rank 0's link
    // rate, rank 0's override.
    const uint32_t H = 8; // header words of a row; then N group sizes (what travels: k-mers or records), then N k-mer counts
    const uint32_t RW = H + 2 * N;
    std::vector<uint64_t> mine(RW), all((size_t) RW * N);
    mine[0] = n_s;
    mine[1] = h_sn[1] ? 0 : d_s;
    mine[2] = n_local;
    mine[3] = c->unmerged ? 1 : 0;
    // (a table that is not allocated yet -- KMU_COUNT_HINT_OCCURRENCES -- enters the model with the size it is going to get)
    mine[4] = c->deferred ? 8ull * (uint64_t) (1.5 * (double) n_local / (n_s && d_s ? std::max(1.0, (double) n_s / d_s) : 1.0)) : (uint64_t) table_image_bytes(c);
    mine[5] = (uint64_t) (local_xgmi_gbps() * 1000.0);
    mine[6] = 0;
    if (const char *e = getenv("KMU_COUNT_ROUTE")) {
        if (!strcmp(e, "occurrences") || !strcmp(e, "superkmers")) mine[6] = KMU_ROUTE_OCCURRENCES;
        if (!strcmp(e, "merge")) mine[6] = KMU_ROUTE_MERGE;
    }
    mine[7] = (uint64_t) c->okind;
    for (uint32_t p = 0; p < N; p++) {
        mine[H + p] = bounds[p + 1] - bounds[p];
        mine[H + N + p] = kto[p];
    }
    KMU_TRY(comm_allgather_host(ctx, mine.data(), all.data(), (uint64_t) RW * 8));
    double sum_ns = 0, sum_ds = 0, sum_n = 0, sum_g = 0, max_table = 0;
    bool valid = true;
    for (uint32_t r = 0; r < N; r++) {
        const uint64_t *row = &all[(size_t) r * RW];
        sum_ns += (double) row[0];
        sum_ds += (double) row[1];
        sum_n += (double) row[2];
        max_table = std::max(max_table, (double) row[4]);
        if (row[0] && !row[1]) valid = false; // a truncated sample somewhere
        if (row[7] != (uint64_t) c->okind)
            return fail(ctx, KMU_E_BAD_ARG, "rank %u counts with another owner function than rank %d (KMU_COUNT_OWNER_HASH / KMU_COUNT_OWNER must agree)", r, cm->rank);
        for (uint32_t p = 0; p < N; p++) sum_g += (double) row[H + p];
    }
    const double ratio = valid && sum_ds > 0 ? sum_ns / sum_ds : 0.0;
    const double n_loc = sum_n / N, d_loc = ratio > 0 ? n_loc / ratio : n_loc;
    // the table of a counter created with KMU_COUNT_HINT_OCCURRENCES: every rank ends up owning about 1 / N of the job's distinct
    // k-mers (the MERGE route holds its shard's own first: no more than that many either); the same size on every rank
    KMU_TRY(table_alloc_for(c, nullptr, 0, d_err, ratio > 0 ? ratio : 1.0, (uint64_t) n_loc + 1));
    double ms_occ, ms_merge;
    if (smer) route_model_smer(max_table, (double) all[5] / 1000.0, n_loc, sum_g / N, d_loc, (int) N, &ms_occ, &ms_merge);
    else route_model(max_table, (double) all[5] / 1000.0, n_loc, d_loc, (int) N, &ms_occ, &ms_merge);
    int route = ms_merge < ms_occ ? KMU_ROUTE_MERGE : KMU_ROUTE_OCCURRENCES;
    if (all[6]) route = (int) all[6]; // KMU_COUNT_ROUTE of rank 0's process
    if (smer && route == KMU_ROUTE_OCCURRENCES) route = KMU_ROUTE_SUPERKMERS; // (the occurrences travel, as records)
    const double f = N > 1 ? (double) (N - 1) / N : 0.0;
    const uint64_t g_local = bounds[N], g_self = mine[H + cm->rank];
    cm->stats.route = route;
    cm->stats.sample_shift = (int32_t) shift;
    cm->stats.dup_ratio = ratio;
    cm->stats.kmers_local = n_local;
    cm->stats.records_local = smer ? g_local : 0;
    cm->stats.bytes_occurrences = (uint64_t) ((smer ? 12.0 : 8.0) * (double) (g_local - g_self));
    cm->stats.bytes_merge = (uint64_t) (12.0 * f * (ratio > 0 ? (double) n_local / ratio : (double) n_local));
    cm->stats.model_ms_occurrences = ms_occ;
    cm->stats.model_ms_merge = ms_merge;
    c->pending = false;
    if (route == KMU_ROUTE_MERGE) {
        if (total_bases && ds.n_seq) KMU_TRY(local_add(c, ds, total_bases, d_err));
        c->unmerged = true;
        return KMU_OK;
    }
    // ---- OCCURRENCES / SUPERKMERS: group, exchange ----
    const uint32_t eb = smer ? SMER_REC_BYTES : 8u;
    void *grouped = nullptr;
    if (smer) {
        KMU_TRY(dev_buf(ctx, "cnt.smer_send", (size_t) g_local * eb + 64, &grouped));
        KMU_TRY(smer_scatter(ctx, ds, total_bases, sg, grouped));
    } else {
        uint64_t *g8 = nullptr;
        KMU_TRY(owner_scatter(c, ds, total_bases, op, &g8));
        grouped = g8;
    }
    std::vector<uint64_t> scnt(N), sdis(N), rcnt(N), rdis(N);
    uint64_t n_recv = 0, k_recv = 0;
    for (uint32_t p = 0; p < N; p++) {
        scnt[p] = bounds[p + 1] - bounds[p];
        sdis[p] = bounds[p];
        rcnt[p] = all[(size_t) p * RW + H + cm->rank];
        rdis[p] = n_recv;
        n_recv += rcnt[p];
        k_recv += all[(size_t) p * RW + H + N + cm->rank];
    }
    void *recv;
    KMU_TRY(dev_buf(ctx, "cnt.recv", n_recv * eb + 64, &recv));
    KMU_HIP(ctx, hipEventRecord(cm->ev_a, ctx->stream));
    KMU_HIP(ctx, hipStreamWaitEvent(cm->stream, cm->ev_a, 0));
    KMU_TRY(comm_alltoallv(ctx, grouped, scnt.data(), sdis.data(), recv, rcnt.data(), rdis.data(), eb, cm->stream));
    KMU_HIP(ctx, hipEventRecord(cm->ev_b, cm->stream));
    c->pend_recv = n_recv;
    c->pend_kmers = k_recv;
    c->pending = true;
    return KMU_OK;
}

// second half: the owner builds its table from what arrived
static int dist_add_end(kmu_counter *c) {
    kmu_ctx *ctx = c->ctx;
    if (!c->pending) return KMU_OK;
    c->pending = false;
    if (!ctx->comm) return fail(ctx, KMU_E_BAD_ARG, "the communicator went away under an exchange of this counter (kmu_comm_destroy)");
    KMU_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->comm->ev_b, 0));
    if (c->pend_recv == 0) return KMU_OK;
    if (c->okind == 1) return add_superkmers(c, ctx->bufs["cnt.recv"].p, c->pend_recv, c->pend_kmers);
    return add_entries(c, (const uint64_t *) ctx->bufs["cnt.recv"].p, nullptr, c->pend_recv, KMU_MEM_DEVICE);
}

extern "C" {

int kmu_count_reset(kmu_counter *c) {
    if (!c) return KMU_E_BAD_ARG;
    if (c->pending && c->ctx->comm) // an exchange left open: what it writes ("cnt.recv") is dropped, but not under later work
        (void) hipStreamWaitEvent(c->ctx->stream, c->ctx->comm->ev_b, 0);
    c->pending = false;
    c->pend_recv = 0;
    c->pend_kmers = 0;
    c->unmerged = false;
    c->empty = true; // materialised lazily: a partitioned build writes every region itself
    c->compact = false;
    c->stats_cached = false;
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
    hipError_t e0 = hipMalloc((void **) &c->scalars, 64);
    if (e0 != hipSuccess) {
        delete c;
        (void) hipGetLastError();
        return fail(ctx, KMU_E_OOM, "cannot allocate a counter");
    }
    if (p->flags & KMU_COUNT_HINT_OCCURRENCES) c->deferred = true; // the first add sizes the table (table_alloc_for)
    else {
        const int rc = table_alloc(c, p->capacity_hint);
        if (rc != KMU_OK) {
            (void) hipFree(c->scalars);
            delete c;
            return rc;
        }
    }
    c->empty = true;
    if (p->flags & KMU_COUNT_DISTRIBUTED) {
        if (!ctx->comm) {
            kmu_count_destroy(c);
            return fail(ctx, KMU_E_BAD_ARG, "KMU_COUNT_DISTRIBUTED needs a communicator on the context (kmu_comm_init)");
        }
        c->dist = true;
        // who owns a k-mer: its minimizer wherever a window fits (Kmer64bit, k >= 17: the occurrences then travel as super-k-mer
        // records, ~1.35 B per k-mer instead of 8), else -- or with KMU_COUNT_OWNER_HASH / KMU_COUNT_OWNER=hash -- the reference's
        // dispatch.  Every rank of a pool must choose alike (the choice is a function of the parameters and the environment).
        c->okind = smer_supported(p->kmer_type, p->kmer_size) && !(p->flags & KMU_COUNT_OWNER_HASH) ? 1 : 0;
        if (const char *e = getenv("KMU_COUNT_OWNER")) {
            if (!strcmp(e, "hash")) c->okind = 0;
            if (!strcmp(e, "minimizer") && smer_supported(p->kmer_type, p->kmer_size)) c->okind = 1;
        }
    }
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
    if (c->rcount) (void) hipFree(c->rcount);
    delete c;
}

namespace kmu {
__global__ void __launch_bounds__(256) k_rebase_offsets(const uint64_t *in, uint64_t n, uint64_t shift, uint64_t *out) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = in[i] - shift;
}
}

// Length of the flat base stream the (unpacked) sequences occupy.  The sequences may be a range of a larger read set
// (offsets[0] > 0, e.g. `offsets + first` of device-resident reads): host input was staged re-based to 0; for device input
// the base pointer is moved forward by whole wave-steps and a re-based copy of the offsets is used, so that neither the
// scratch sizes nor the walk depend on what lies before the range (the kernels skip the few bases left before offsets[0]).
static int flat_stream_extent(kmu_ctx *ctx, const uint64_t *host_offsets, uint32_t n_seq, int mem, DevSeqs &ds, uint64_t *total_out) {
    if (mem == KMU_MEM_HOST) {
        *total_out = n_seq ? host_offsets[n_seq] - host_offsets[0] : 0;
        return KMU_OK;
    }
    uint64_t first = 0, last = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&first, ds.offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipMemcpyAsync(&last, ds.offsets + n_seq, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t shift = first & ~(uint64_t) 1023;
    if (shift) {
        void *q;
        KMU_TRY(dev_buf(ctx, "in.offsets", ((size_t) n_seq + 1) * 8, &q));
        const int grid = (int) std::min<uint64_t>(((uint64_t) n_seq + 256) / 256, (uint64_t) ctx->num_cus * 8);
        hipLaunchKernelGGL(k_rebase_offsets, dim3(grid), dim3(256), 0, ctx->stream, ds.offsets, (uint64_t) n_seq + 1, shift, (uint64_t *) q);
        ds.bases += shift;
        ds.offsets = (const uint64_t *) q;
    }
    *total_out = last - shift;
    return KMU_OK;
}

} // extern "C"

namespace kmu {
// kmu_sketch_count on host buffers: the count's level-1 partition runs chunk by chunk under the upload (the single-pass
// partition needs no histogram of the whole batch).  count_chunked_begin returns *on = 0 when that route does not apply
// (a distributed counter, a small batch, a one-level table): the caller then adds the reads in one go at the end.
struct CountChunked {
    SegRun run;
};
int count_chunked_begin(kmu_counter *c, DevSeqs &all, const uint64_t *host_offsets, uint32_t *d_err, void **handle, int *on) {
    *on = 0;
    *handle = nullptr;
    if (c->dist || all.n_seq == 0) return KMU_OK;
    uint64_t total_bases = 0;
    KMU_TRY(flat_stream_extent(c->ctx, host_offsets, all.n_seq, KMU_MEM_HOST, all, &total_bases));
    KMU_TRY(table_alloc_for(c, nullptr, 0, d_err)); // (the reads are not on the device yet: a table by the hint)
    PartPlan pl;
    const bool partitioned = partitioned_batch_wanted(c, total_bases);
    if (!partitioned || !part_plan_for(c, &pl) || !pl.b2 || !seg_partition_wanted(total_bases)) return KMU_OK;
    CountChunked *h = new CountChunked();
    const int rc = seg_begin(c, all, total_bases, pl, d_err, &h->run, true);
    if (rc != KMU_OK) { delete h; return rc; }
    *handle = h;
    *on = 1;
    return KMU_OK;
}
int count_chunked_level1(kmu_counter *c, void *handle, uint64_t bases_ready) { return seg_level1(c, &((CountChunked *) handle)->run, bases_ready); }
int count_chunked_finish(kmu_counter *c, void *handle) {
    CountChunked *h = (CountChunked *) handle;
    int taken = 0;
    int rc = seg_finish(c, &h->run, &taken);
    if (rc == KMU_OK && !taken) { // a segment overflowed: the exact route (not a second attempt on the same k-mers)
        c->no_seg = true;
        rc = local_add(c, h->run.ds, h->run.total_bases, h->run.d_err);
        c->no_seg = false;
    }
    delete h;
    return rc;
}
void count_chunked_abort(void *handle) { delete (CountChunked *) handle; }

// What kmu_sketch_count needs of a counter: the canonical k-mers of device-resident unpacked reads go in, in two halves
// (a distributed counter's exchange is in flight between them: the caller's kernels on the context's stream run under it).
// host_offsets: the caller's host copy of the offsets (mem == KMU_MEM_HOST: already re-based to the staged stream) or null.
int count_add_device_begin(kmu_counter *c, DevSeqs &ds, const uint64_t *host_offsets, int mem, uint32_t *d_err) {
    kmu_ctx *ctx = c->ctx;
    if (c->dist && c->pending) KMU_TRY(dist_add_end(c));
    uint64_t total_bases = 0;
    if (ds.n_seq) KMU_TRY(flat_stream_extent(ctx, host_offsets, ds.n_seq, mem, ds, &total_bases));
    if (c->dist) return dist_add_begin(c, ds, total_bases, d_err);
    if (ds.n_seq) return local_add(c, ds, total_bases, d_err);
    return KMU_OK;
}
int count_add_device_end(kmu_counter *c) { return c->dist ? dist_add_end(c) : KMU_OK; }
kmu_ctx *counter_ctx(kmu_counter *c) { return c->ctx; }
} // namespace kmu

extern "C" {

int kmu_count_add_reads(kmu_counter *c, const uint8_t *bases, const uint64_t *offsets, const uint64_t *packed_offsets,
                        uint32_t n_seq, int input_kind, int mem) {
    if (!c) return KMU_E_BAD_ARG;
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, packed_offsets, n_seq, input_kind, mem, &ds));
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    if (c->dist) { // collective: every rank passes here, also with an empty shard
        if (ds.packed) return fail(ctx, KMU_E_UNSUPPORTED, "distributed counting takes unpacked (ASCII) reads");
        if (c->pending) KMU_TRY(dist_add_end(c)); // (an exchange left open by kmu_sketch_count)
        uint64_t total_bases = 0;
        if (n_seq) KMU_TRY(flat_stream_extent(ctx, offsets, n_seq, mem, ds, &total_bases));
        KMU_TRY(dist_add_begin(c, ds, total_bases, d_err));
        KMU_TRY(dist_add_end(c));
        if (!(mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
        return finish_call(ctx, mem);
    }
    if (n_seq) {
        uint64_t total_bases = 0;
        if (!ds.packed) KMU_TRY(flat_stream_extent(ctx, offsets, n_seq, mem, ds, &total_bases));
        KMU_TRY(local_add(c, ds, total_bases, d_err));
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
    KMU_TRY(table_alloc_for(c, nullptr, 0, d_err, 0.0, std::max<uint64_t>(c->p.capacity_hint, n)));
    {
        const char *force = getenv("KMU_COUNT_PATH");
        bool partitioned = !counts && partitioned_batch_wanted(c, n) && c->lg - c->rbits <= 22;
        if (force && !strcmp(force, "direct")) partitioned = false;
        if (force && !strcmp(force, "partitioned")) partitioned = !counts;
        if (partitioned) {
            KMU_TRY(partitioned_add_kmers(c, d_k, n, d_err));
            if (!(mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
            return finish_call(ctx, mem);
        }
    }
    KMU_TRY(materialize(c));
    {
        KernelTimer tm(ctx, "k_count_add_kmers");
        hipLaunchKernelGGL(k_count_add_kmers, dim3(grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, d_k, d_c, n,
                           table_of(c), d_err);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (!(mem == KMU_MEM_DEVICE && ctx->async_device)) KMU_TRY(check_err_word(ctx, d_err));
    return finish_call(ctx, mem);
}

// the k-mers of n_rec super-k-mer records in device memory (kmu_smer.h: what the owner of a key range receives) into this
// table: big batches straight into the single-pass partition (k_smer_scatter1 expands a record inside the tile sort), small
// ones and the fall-back through an array of their canonical k-mers
static int add_superkmers(kmu_counter *c, const void *recs, uint64_t n_rec, uint64_t n_kmers) {
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (n_rec == 0) return KMU_OK;
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    KMU_TRY(table_alloc_for(c, nullptr, 0, d_err, 0.0, std::max<uint64_t>(c->p.capacity_hint, n_kmers)));
    const char *force = getenv("KMU_COUNT_PATH");
    bool partitioned = partitioned_batch_wanted(c, n_kmers) && c->lg - c->rbits <= 22;
    if (force && !strcmp(force, "direct")) partitioned = false;
    if (force && !strcmp(force, "partitioned")) partitioned = true;
    PartPlan pl;
    bool overflowed = false;
    if (partitioned && part_plan_for(c, &pl) && pl.b2 && seg_partition_wanted(n_kmers)) {
        int taken = 0;
        KMU_TRY(seg_partitioned_add_kmers(c, nullptr, n_kmers, pl, d_err, &taken, recs, n_rec));
        if (taken) {
            if (!ctx->async_device) KMU_TRY(check_err_word(ctx, d_err));
            return finish_call(ctx, KMU_MEM_DEVICE);
        }
        overflowed = true; // a segment and the spill list overflowed, the table is untouched: the exact levels, on the k-mers
    }
    void *x, *cur;
    KMU_TRY(dev_buf(ctx, "cnt.smer_x", n_kmers * 8 + 64, &x));
    KMU_TRY(dev_buf(ctx, "cnt.smer_cur", 64, &cur));
    KMU_TRY(smer_expand(ctx, recs, n_rec, c->p.kmer_size, (uint64_t *) x, (uint64_t *) cur));
    c->no_seg = overflowed;
    const int rc = add_entries(c, (const uint64_t *) x, nullptr, n_kmers, KMU_MEM_DEVICE);
    c->no_seg = false;
    return rc;
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
    KMU_TRY(materialize(c));
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

// the largest count a slot of this table holds: what a region build may leave (quotient slots: 2^w - 1024, and 1 024 less where
// the build runs 1 024 threads -- regions of 8 192 slots --: every adder stops there, the plain adds in flight stay below the field)
static uint64_t count_ceiling(const kmu_counter *c) {
    if (!c->qw) return 0xFFFFFFFFull;
    return (1ull << c->qw) - (uint64_t) Q_MARGIN - (c->rbits > 12 ? (uint64_t) Q_MARGIN : 0ull);
}

static int count_stats(kmu_counter *c, uint64_t *distinct, uint64_t *unique, uint64_t *occurrences = nullptr, uint64_t *saturated = nullptr) {
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (c->empty) {
        if (distinct) *distinct = 0;
        if (unique) *unique = 0;
        if (occurrences) *occurrences = 0;
        if (saturated) *saturated = 0;
        return KMU_OK;
    }
    if (!saturated && c->compact && c->stats_cached) { // the last build left them
        uint64_t h[3];
        KMU_HIP(ctx, hipMemcpyAsync(h, c->scalars + 4, 24, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (distinct) *distinct = h[0];
        if (unique) *unique = h[1];
        if (occurrences) *occurrences = h[2];
        return KMU_OK;
    }
    KMU_TRY(materialize(c));
    KMU_HIP(ctx, hipMemsetAsync(c->scalars, 0, 32, ctx->stream));
    {
        KernelTimer tm(ctx, "k_count_stats");
        hipLaunchKernelGGL(k_count_stats, dim3(grid_for(ctx, c->nslots, 1024)), dim3(256), 0, ctx->stream, table_of(c),
                           c->nslots, c->scalars, (uint32_t) std::min<uint64_t>(count_ceiling(c), 0xFFFFFFFFull));
    }
    KMU_HIP(ctx, hipGetLastError());
    uint64_t h[4];
    KMU_HIP(ctx, hipMemcpyAsync(h, c->scalars, 32, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (distinct) *distinct = h[0];
    if (unique) *unique = h[1];
    if (occurrences) *occurrences = h[3];
    if (saturated) *saturated = h[2];
    return KMU_OK;
}

int kmu_count_nb_saturated(kmu_counter *c, uint64_t *out) {
    if (!c || !out) return KMU_E_BAD_ARG;
    return count_stats(c, nullptr, nullptr, nullptr, out);
}

int kmu_count_nb_distinct(kmu_counter *c, uint64_t *out) {
    if (!c || !out) return KMU_E_BAD_ARG;
    return count_stats(c, out, nullptr);
}
int kmu_count_nb_unique(kmu_counter *c, uint64_t *out) {
    if (!c || !out) return KMU_E_BAD_ARG;
    return count_stats(c, nullptr, out);
}
int kmu_count_nb_occurrences(kmu_counter *c, uint64_t *out) {
    if (!c || !out) return KMU_E_BAD_ARG;
    return count_stats(c, nullptr, nullptr, out);
}
int kmu_count_table_info(const kmu_counter *c, kmu_count_table_info_t *out) {
    if (!c || !out) return KMU_E_BAD_ARG;
    out->nslots = c->nslots; // (0 while a KMU_COUNT_HINT_OCCURRENCES counter waits for its first add)
    out->bytes_per_slot = c->qw ? 8u : 12u;
    out->count_field_bits = c->qw ? (uint32_t) c->qw : 32u;
    out->table_bytes = (uint64_t) table_image_bytes(c);
    out->count_ceiling = count_ceiling(c);
    return KMU_OK;
}

// shared by dump / export: select into device buffers, then hand over
static int select_entries(kmu_counter *c, uint32_t min_count, uint32_t maxc, uint32_t part, uint32_t n_parts,
                          uint64_t *kmers_out, uint32_t *counts_out, uint64_t cap, int mem, bool sort, uint64_t *n_out,
                          bool own_owner = false /* parts by the counter's own owner function instead of the reference's dispatch */) {
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (c->empty) {
        *n_out = 0;
        return KMU_OK;
    }
    KMU_TRY(materialize(c));
    const int w32 = own_owner ? owner_mode_of(c) : kmer_val_bytes(c->p.kmer_type) == 4;
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
    KMU_HIP(ctx, hipMemsetAsync(c->scalars, 0, 32, ctx->stream));
    {
        KernelTimer tm(ctx, "k_count_select");
        hipLaunchKernelGGL(k_count_select, dim3(grid_for(ctx, c->nslots, 1024)), dim3(256), 0, ctx->stream, table_of(c),
                           c->nslots, min_count, maxc, w32, part, n_parts, cap, d_k, d_c, c->scalars + 2);
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
    return select_entries(c, min_count, max_count(c), 0, 0, kmers_out, counts_out, cap, KMU_MEM_HOST, true, n_out);
}

int kmu_count_export_part(kmu_counter *c, uint32_t part, uint32_t n_parts, uint64_t *kmers_out, uint32_t *counts_out,
                          uint64_t cap, int mem, uint64_t *n_out) {
    if (!c || !n_out || n_parts == 0 || part >= n_parts || (kmers_out && !counts_out)) return KMU_E_BAD_ARG;
    // raw (unclamped) counts are exported so that merged totals saturate only once, at query time
    return select_entries(c, 1u, 0xFFFFFFFFu, part, n_parts, kmers_out, counts_out, cap, mem, false, n_out);
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

// KmerCounter::eliminate_once_kmer (kmercount.rs:110-117): the singletons are dropped, counts >= 2 stay
int kmu_count_eliminate_once(kmu_counter *c) {
    if (!c) return KMU_E_BAD_ARG;
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t n = 0;
    KMU_TRY(select_entries(c, 2u, 0xFFFFFFFFu, 0, 0, nullptr, nullptr, 0, KMU_MEM_DEVICE, false, &n));
    void *k = nullptr, *cc = nullptr;
    KMU_TRY(dev_buf(ctx, "cnt.keep.k", n * 8 + 8, &k));
    KMU_TRY(dev_buf(ctx, "cnt.keep.c", n * 4 + 8, &cc));
    uint64_t n2 = 0;
    KMU_TRY(select_entries(c, 2u, 0xFFFFFFFFu, 0, 0, (uint64_t *) k, (uint32_t *) cc, n, KMU_MEM_DEVICE, false, &n2));
    KMU_TRY(kmu_count_reset(c));
    int rc = add_entries(c, (const uint64_t *) k, (const uint32_t *) cc, n2, KMU_MEM_DEVICE);
    if (rc) return rc;
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KMU_OK;
}

int kmu_count_once_positions(kmu_counter *c, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int mem,
                             uint64_t *kmers_out, uint32_t *numseq_out, uint32_t *numkmer_out, uint64_t cap, uint64_t *n_out) {
    if (!c || !n_out || (kmers_out && (!numseq_out || !numkmer_out))) return KMU_E_BAD_ARG;
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    *n_out = 0;
    if (n_seq == 0) return KMU_OK;
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, nullptr, n_seq, KMU_INPUT_ASCII, mem, &ds));
    uint64_t total_bases = 0;
    KMU_TRY(flat_stream_extent(ctx, offsets, n_seq, mem, ds, &total_bases));
    if (total_bases == 0) return KMU_OK;
    KMU_TRY(materialize(c));
    const uint64_t nsteps = ((total_bases + 15) / 16 + 63) / 64;
    void *cnt, *base;
    KMU_TRY(dev_buf(ctx, "once.cnt", nsteps * 4 + 64, &cnt));
    KMU_TRY(dev_buf(ctx, "once.base", (nsteps + 1) * 8 + 64, &base));
    const int grid = (int) std::min<uint64_t>((nsteps + 3) / 4, (uint64_t) ctx->num_cus * 8);
    {
        KernelTimer tm(ctx, "k_once_count");
        hipLaunchKernelGGL(k_once_count, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets, n_seq, c->p.kmer_size,
                           table_of(c), (uint32_t *) cnt);
    }
    KMU_TRY(device_scan_u32(ctx, (const uint32_t *) cnt, nsteps, (uint64_t *) base));
    uint64_t n = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&n, (const uint64_t *) base + nsteps, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    if (!kmers_out) return KMU_OK; // size query
    if (cap < n) return fail(ctx, KMU_E_BAD_ARG, "output too small: %llu records", (unsigned long long) n);
    if (n == 0) return KMU_OK;
    uint64_t *d_k = kmers_out;
    uint32_t *d_s = numseq_out, *d_p = numkmer_out;
    if (mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "once.k", n * 8 + 64, &q));
        d_k = (uint64_t *) q;
        KMU_TRY(dev_buf(ctx, "once.s", n * 4 + 64, &q));
        d_s = (uint32_t *) q;
        KMU_TRY(dev_buf(ctx, "once.p", n * 4 + 64, &q));
        d_p = (uint32_t *) q;
    }
    {
        KernelTimer tm(ctx, "k_once_emit");
        hipLaunchKernelGGL(k_once_emit, dim3(grid), dim3(256), 0, ctx->stream, ds.bases, ds.offsets, n_seq, c->p.kmer_size,
                           table_of(c), (const uint64_t *) base, d_k, d_s, d_p);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (mem == KMU_MEM_HOST) {
        KMU_HIP(ctx, hipMemcpyAsync(kmers_out, d_k, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipMemcpyAsync(numseq_out, d_s, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipMemcpyAsync(numkmer_out, d_p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    return finish_call(ctx, mem);
}

// Collective.  After it the counter of rank r holds the k-mers with owner r and their multiplicities over all ranks.
int kmu_count_finalize(kmu_counter *c) {
    if (!c) return KMU_E_BAD_ARG;
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (!c->dist) return KMU_OK; // a counter of one rank is its own pool
    kmu_comm *cm = ctx->comm;
    if (!cm) return fail(ctx, KMU_E_BAD_ARG, "the communicator of this counter's context is gone");
    KMU_TRY(dist_add_end(c));
    const uint32_t N = (uint32_t) cm->nranks, me = (uint32_t) cm->rank;
    const int w32 = owner_mode_of(c);
    // who has entries of other owners, and how many for whom
    void *po;
    KMU_TRY(dev_buf(ctx, "cnt.per_owner", ((size_t) N + 1) * 8 * 2 + 64, &po));
    KMU_HIP(ctx, hipMemsetAsync(po, 0, ((size_t) N + 1) * 8 * 2, ctx->stream));
    const bool have = c->unmerged && !c->empty;
    if (have) KMU_TRY(materialize(c)); // (the export probes and rewrites slots)
    if (have) {
        KernelTimer tm(ctx, "k_owner_census");
        hipLaunchKernelGGL(k_owner_census, dim3(grid_for(ctx, c->nslots, 1024)), dim3(256), ((size_t) N + 1) * 4, ctx->stream, table_of(c),
                           c->nslots, w32, N, (unsigned long long *) po);
    }
    std::vector<uint64_t> mine(N + 1), all((size_t) (N + 1) * N);
    KMU_HIP(ctx, hipMemcpyAsync(mine.data(), po, ((size_t) N + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t n_stay = mine[me]; // entries this rank owns itself: they stay
    const uint64_t n_tomb = mine[N];  // slots still held by entries that left at EARLIER finalizes (they lengthen probe chains like live ones)
    mine[me] = 0;
    mine[N] = c->unmerged ? 1 : 0;
    KMU_TRY(comm_allgather_host(ctx, mine.data(), all.data(), ((uint64_t) N + 1) * 8));
    bool any = false;
    for (uint32_t r = 0; r < N; r++) any |= all[(size_t) r * (N + 1) + N] != 0;
    c->unmerged = false;
    if (!any) return KMU_OK;
    std::vector<uint64_t> scnt(N), sdis(N), rcnt(N), rdis(N);
    uint64_t n_send = 0, n_recv = 0;
    for (uint32_t p = 0; p < N; p++) {
        scnt[p] = mine[p];
        sdis[p] = n_send;
        n_send += scnt[p];
        rcnt[p] = all[(size_t) p * (N + 1) + me];
        rdis[p] = n_recv;
        n_recv += rcnt[p];
    }
    void *sk, *sc, *rk, *rc;
    KMU_TRY(dev_buf(ctx, "cnt.send_k", n_send * 8 + 64, &sk));
    KMU_TRY(dev_buf(ctx, "cnt.send_c", n_send * 4 + 64, &sc));
    KMU_TRY(dev_buf(ctx, "cnt.recv", n_recv * 8 + 64, &rk));
    KMU_TRY(dev_buf(ctx, "cnt.recv_c", n_recv * 4 + 64, &rc));
    if (n_send) {
        unsigned long long *cursor = (unsigned long long *) po + (N + 1);
        KMU_HIP(ctx, hipMemcpyAsync(cursor, sdis.data(), (size_t) N * 8, hipMemcpyHostToDevice, ctx->stream));
        KernelTimer tm(ctx, "k_owner_emit");
        hipLaunchKernelGGL(k_owner_emit, dim3(grid_for(ctx, c->nslots, 1024)), dim3(256), 0, ctx->stream, table_of(c), c->nslots, w32, me,
                           N, cursor, (uint64_t *) sk, (uint32_t *) sc);
        KMU_HIP(ctx, hipGetLastError());
    }
    const uint64_t sent0 = cm->stats.bytes_sent;
    KMU_HIP(ctx, hipEventRecord(cm->ev_a, ctx->stream));
    KMU_HIP(ctx, hipStreamWaitEvent(cm->stream, cm->ev_a, 0));
    KMU_TRY(comm_alltoallv(ctx, sk, scnt.data(), sdis.data(), rk, rcnt.data(), rdis.data(), 8, cm->stream));
    KMU_TRY(comm_alltoallv(ctx, sc, scnt.data(), sdis.data(), rc, rcnt.data(), rdis.data(), 4, cm->stream));
    KMU_HIP(ctx, hipEventRecord(cm->ev_b, cm->stream));
    KMU_HIP(ctx, hipStreamWaitEvent(ctx->stream, cm->ev_b, 0));
    cm->stats.bytes_merge = cm->stats.bytes_sent - sent0; // exact now
    // The entries that left are still in the table as tombstones of their probe chains (key kept, count zero).  Where they
    // and what arrives would crowd the table (a rank whose shard holds as many distinct k-mers as it ends up owning: every
    // k-mer of a noisy long-read set occurs once), the table is rebuilt from the entries that stay before the merge.
    // (a local decision: every rank looks at its own table, nothing collective follows from it)
    if ((n_send || n_tomb) && (double) (n_stay + n_send + n_tomb + n_recv) > 0.55 * (double) c->nslots) {
        void *kk = nullptr, *kc = nullptr;
        KMU_TRY(dev_buf(ctx, "cnt.keep.k", n_stay * 8 + 8, &kk));
        KMU_TRY(dev_buf(ctx, "cnt.keep.c", n_stay * 4 + 8, &kc));
        uint64_t n2 = 0;
        KMU_TRY(select_entries(c, 1u, 0xFFFFFFFFu, me, N, (uint64_t *) kk, (uint32_t *) kc, n_stay, KMU_MEM_DEVICE, false, &n2, true));
        KMU_TRY(kmu_count_reset(c));
        if (n2) KMU_TRY(add_entries(c, (const uint64_t *) kk, (const uint32_t *) kc, n2, KMU_MEM_DEVICE));
    }
    if (n_recv) KMU_TRY(add_entries(c, (const uint64_t *) rk, (const uint32_t *) rc, n_recv, KMU_MEM_DEVICE));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream)); // sdis / scnt are locals the copies above read
    return KMU_OK;
}

// host copies of the Wang hashes (kmu_device.h): DispatchableT::dispatch, kmercount.rs:382-420
static uint32_t host_int32_hash(uint32_t key) {
    key = ~key + (key << 15);
    key = key ^ (key >> 12);
    key = key + (key << 2);
    key = key ^ (key >> 4);
    key = key * 2057u;
    key = key ^ (key >> 16);
    return key;
}
static uint64_t host_int64_hash(uint64_t key) {
    key = ~key + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}
int kmu_kmer_owner(int kmer_type, const uint64_t *canon_kmers, uint64_t n, uint32_t n_parts, uint32_t *owners_out) {
    if ((!canon_kmers || !owners_out) && n) return KMU_E_BAD_ARG;
    if (n_parts == 0) return KMU_E_BAD_ARG;
    const bool w32 = kmer_val_bytes(kmer_type) == 4;
    for (uint64_t i = 0; i < n; i++)
        owners_out[i] = w32 ? host_int32_hash((uint32_t) canon_kmers[i]) % n_parts : (uint32_t) (host_int64_hash(canon_kmers[i]) % n_parts);
    return KMU_OK;
}

int kmu_count_extract_by_owner(kmu_counter *c, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int mem,
                                uint32_t n_parts, uint64_t **dev_kmers_out, uint64_t *part_bounds_out) {
    if (!c || !dev_kmers_out || !part_bounds_out) return KMU_E_BAD_ARG;
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, nullptr, n_seq, KMU_INPUT_ASCII, mem, &ds));
    uint64_t total_bases = 0;
    KMU_TRY(flat_stream_extent(ctx, offsets, n_seq, mem, ds, &total_bases));
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    KMU_TRY(extract_by_owner(c, ds, total_bases, n_parts, dev_kmers_out, part_bounds_out, d_err));
    KMU_TRY(check_err_word(ctx, d_err));
    if (ctx->profiling) profile_collect(ctx);
    return KMU_OK;
}

int kmu_count_owner_kind(const kmu_counter *c) { return c && c->dist ? c->okind : KMU_OWNER_HASH; }

int kmu_count_extract_superkmers(kmu_counter *c, const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int mem, uint32_t n_parts,
                                 void **dev_records_out, uint64_t *record_bounds_out, uint64_t *kmers_per_part_out) {
    if (!c || !dev_records_out || !record_bounds_out) return KMU_E_BAD_ARG;
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (!smer_supported(c->p.kmer_type, c->p.kmer_size)) return fail(ctx, KMU_E_UNSUPPORTED, "super-k-mers need Kmer64bit with 17 <= k <= 31");
    DevSeqs ds;
    KMU_TRY(stage_sequences(ctx, bases, offsets, nullptr, n_seq, KMU_INPUT_ASCII, mem, &ds));
    uint64_t total_bases = 0;
    KMU_TRY(flat_stream_extent(ctx, offsets, n_seq, mem, ds, &total_bases));
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    SmerGroups sg;
    KMU_TRY(smer_census(ctx, ds, total_bases, c->p.kmer_size, n_parts, d_err, SampleArgs{nullptr, nullptr, 0u, 0u}, &sg));
    KMU_HIP(ctx, hipMemcpyAsync(record_bounds_out, sg.binstart, (size_t) (n_parts + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (kmers_per_part_out) KMU_HIP(ctx, hipMemcpyAsync(kmers_per_part_out, sg.kmers, (size_t) n_parts * 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    void *out;
    KMU_TRY(dev_buf(ctx, "cnt.smer_send", (size_t) record_bounds_out[n_parts] * SMER_REC_BYTES + 64, &out));
    KMU_TRY(smer_scatter(ctx, ds, total_bases, sg, out));
    KMU_TRY(check_err_word(ctx, d_err));
    if (ctx->profiling) profile_collect(ctx);
    *dev_records_out = out;
    return KMU_OK;
}

int kmu_count_add_superkmers(kmu_counter *c, const void *records, uint64_t n_records, int mem) {
    if (!c || (!records && n_records)) return KMU_E_BAD_ARG;
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (!smer_supported(c->p.kmer_type, c->p.kmer_size)) return fail(ctx, KMU_E_UNSUPPORTED, "super-k-mers need Kmer64bit with 17 <= k <= 31");
    if (n_records == 0) return KMU_OK;
    const void *d_r = records;
    if (mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "cnt.in.k", n_records * SMER_REC_BYTES + 64, &q));
        KMU_HIP(ctx, hipMemcpyAsync(q, records, n_records * SMER_REC_BYTES, hipMemcpyHostToDevice, ctx->stream));
        d_r = q;
    }
    void *cur;
    KMU_TRY(dev_buf(ctx, "cnt.smer_cur", 64, &cur));
    uint64_t n_kmers = 0;
    KMU_TRY(smer_count_kmers(ctx, d_r, n_records, (uint64_t *) cur));
    KMU_HIP(ctx, hipMemcpyAsync(&n_kmers, cur, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    KMU_TRY(add_superkmers(c, d_r, n_records, n_kmers));
    return finish_call(ctx, mem);
}

} // extern "C"
