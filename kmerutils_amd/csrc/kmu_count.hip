// kmu_count.hip -- k-mer counting on gfx950 behind the KmerCountT contract (src/base/kmercount.rs:48-59, 241-287): the table and
// its API.  The table format is in kmu_count_table.h, the partitioned build of big batches in kmu_count_part.hip, the
// distributed counter in kmu_count_dist.hip.
//
// Two ways to add k-mers:
//  * big batches (the throughput path, kmu_count_part.hip): a radix-partitioned build -- HBM sees only streaming reads and
//    writes: bases, the partition streams, the table image.
//  * small batches / single values / merges (here): direct insertion with one 64-bit CAS (+ one 32-bit atomic add, wide slots).
#include <algorithm>
#include <cmath>
#include <vector>

#include "kmu_count_table.h"
#include "kmu_flat.h"
#include "kmu_stream.h"

namespace kmu {

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

// scalars[0] += distinct, [1] += unique, [2] += slots whose count sits at (or above) `ceiling`, [3] += occurrences held
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

__global__ void __launch_bounds__(256) k_rebase_offsets(const uint64_t *in, uint64_t n, uint64_t shift, uint64_t *out) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = in[i] - shift;
}

// ------------------------------------------------------------------------------------------------
// the table of a counter
// ------------------------------------------------------------------------------------------------
// for `distinct_hint` distinct canonical k-mers: whole regions of 4096 slots at a load of 0.70 at the hint (KMU_COUNT_LOAD, per
// cent: A/B runs); KMU_COUNT_REGIONS=<n> (tests) forces the region count, whatever the hint.  More than 2048 regions are 2^b1
// groups of n2 each (the two levels of the partitioned build), n2 a multiple of 4.
int table_alloc(kmu_counter *c, uint64_t distinct_hint) {
    kmu_ctx *ctx = c->ctx;
    const kmu_count_params *p = &c->p;
    double load = 0.70;
    if (const char *e = getenv("KMU_COUNT_LOAD")) load = std::min(0.90, std::max(0.10, atof(e) / 100.0));
    const uint64_t want = std::max<uint64_t>(1024, (uint64_t) std::ceil((double) distinct_hint / load));
    uint64_t n_regions;
    if (want <= (1ull << REGION_BITS_MAX)) { // one region, a power of two of slots
        c->rbits = 10;
        while ((1ull << c->rbits) < want) c->rbits++;
        n_regions = 1;
    } else {
        c->rbits = REGION_BITS_MAX;
        n_regions = (want + (1ull << REGION_BITS_MAX) - 1) >> REGION_BITS_MAX;
    }
    // slot format: quotient (8 bytes per slot) whenever the region index leaves room for the count field: w >= 11 for 8-bit
    // counters, >= 17 for 16-bit ones.  KMU_COUNT_FMT=wide keeps 12 bytes per slot; =quot (tests) raises a small table to the
    // size the quotient format starts at.
    const int w_need = p->counter_bits == 8 ? 11 : 17;
    bool force_quot = false, force_wide = false;
    if (const char *e = getenv("KMU_COUNT_FMT")) {
        force_wide = !strcmp(e, "wide");
        force_quot = !strcmp(e, "quot");
    }
    if (const char *e = getenv("KMU_COUNT_REGIONS")) {
        const long long v = atoll(e);
        if (v >= 1) { n_regions = (uint64_t) v; c->rbits = REGION_BITS_MAX; }
    } else if (force_quot && n_regions < (1ull << w_need)) { n_regions = 1ull << w_need; c->rbits = REGION_BITS_MAX; }
    if (n_regions > (uint64_t) GROUP_REGIONS_MAX * GROUP_REGIONS_MAX)
        return fail(ctx, KMU_E_OOM, "a count table of %llu regions of 4096 slots is beyond what one GPU holds", (unsigned long long) n_regions);
    if (n_regions <= GROUP_REGIONS_MAX) {
        c->b1 = 0;
        c->n2 = (uint32_t) n_regions;
    } else {
        int bits = 0;
        while ((1ull << bits) < n_regions) bits++;
        c->b1 = (bits + 1) / 2;
        for (;;) {
            const uint64_t n2 = (((n_regions + (1ull << c->b1) - 1) >> c->b1) + 3) & ~(uint64_t) 3;
            if (n2 <= GROUP_REGIONS_MAX) { c->n2 = (uint32_t) n2; break; }
            c->b1++;
        }
    }
    n_regions = ((uint64_t) c->n2) << c->b1;
    int f = 0;
    while ((2u << f) <= c->n2) f++; // floor(log2 n2)
    const int w = c->b1 + f;
    c->qw = (w >= w_need && !force_wide && c->rbits == REGION_BITS_MAX) ? w : 0;
    c->nslots = n_regions << c->rbits;
    hipError_t e1 = hipMalloc((void **) &c->keys, c->nslots * 8);
    hipError_t e2 = e1 == hipSuccess && !c->qw ? hipMalloc((void **) &c->counts, c->nslots * 4) : e1;
    if (e2 == hipSuccess) e2 = hipMalloc((void **) &c->lox, ((size_t) c->n2 + 1) * 4);
    if (e2 != hipSuccess) {
        if (c->keys) (void) hipFree(c->keys);
        if (c->counts) (void) hipFree(c->counts);
        if (c->lox) (void) hipFree(c->lox);
        c->keys = nullptr;
        c->counts = nullptr;
        c->lox = nullptr;
        const unsigned long long ns = c->nslots;
        c->nslots = 0;
        (void) hipGetLastError();
        return fail(ctx, KMU_E_OOM, "cannot allocate a %llu-slot count table", ns);
    }
    std::vector<uint32_t> lox((size_t) c->n2 + 1);
    for (uint32_t s = 0; s < c->n2; s++) lox[s] = (uint32_t) ((((uint64_t) s << 32) + c->n2 - 1) / c->n2); // ceil(s 2^32 / n2)
    lox[c->n2] = 0xFFFFFFFFu;
    KMU_HIP(ctx, hipMemcpy(c->lox, lox.data(), lox.size() * 4, hipMemcpyHostToDevice));
    c->empty = true;
    c->deferred = false;
    return KMU_OK;
}

// KMU_COUNT_HINT_OCCURRENCES: the table is allocated when the first k-mers come.  `ds` (unpacked reads, or null): their
// duplication is measured first -- a sample by key of the batch (every occurrence of a sampled k-mer is in it), distinct keys
// of the sample counted in a scratch table (sample_ratio, kmu_count_part.hip) -- and the table is sized for occurrences /
// ratio distinct k-mers instead of for the occurrences: config 4's shard (0.75 G occurrences of 0.207 G distinct 31-mers) gets
// a quarter of the slots, all of them written by every build.  One pass over the reads, once per counter.  `known_ratio` > 0:
// the caller has measured it (a distributed add); `occurrences`: the k-mers this table is to hold (0: the hint of kmu_count_create).
int table_alloc_for(kmu_counter *c, const DevSeqs *ds, uint64_t total_bases, uint32_t *d_err, double known_ratio, uint64_t occurrences) {
    if (!c->deferred) return KMU_OK;
    double ratio = known_ratio;
    if (ratio <= 0.0 && ds && !ds->packed && total_bases >= (1u << 16)) KMU_TRY(sample_ratio(c, *ds, total_bases, d_err, &ratio));
    if (const char *e = getenv("KMU_COUNT_DUP_RATIO")) ratio = atof(e); // (tests: the decision without the data)
    const uint64_t occ = occurrences ? occurrences : std::max<uint64_t>(c->p.capacity_hint, ds && !ds->packed ? total_bases : 0);
    uint64_t distinct = ratio > 1.0 ? (uint64_t) ((double) occ / ratio * 1.05) + 1024 : occ; // (5 %: the sample's error)
    return table_alloc(c, std::min(distinct, occ));
}

// make the table's image physical before anything probes or updates slots in place: the logical "all free" state is written out
int materialize(kmu_counter *c) {
    if (c->deferred) KMU_TRY(table_alloc(c, c->p.capacity_hint)); // (a reader before the first add: the table the hint asks for, empty)
    kmu_ctx *ctx = c->ctx;
    if (c->empty) {
        KMU_HIP(ctx, hipMemsetAsync(c->keys, 0xFF, c->nslots * 8, ctx->stream));
        if (!c->qw) KMU_HIP(ctx, hipMemsetAsync(c->counts, 0, c->nslots * 4, ctx->stream));
        c->empty = false;
    }
    return KMU_OK;
}

// Is a batch of n k-mers worth the streaming (partitioned) build?  It rewrites the table image, direct insertion costs ~52 ps per
// k-mer whatever the table: at the bench table of round 3 (2^33 slots, scripts/r03_incr.py) a batch of 1/8 of the slots takes 37.8 ms
// partitioned against 67.6 direct, one of 1/16 33.0 against 28.3; into an EMPTY table (no image to read, and direct insertion has to
// wipe it first) 19.2 against 36.9 at 1/16 of the slots, 16.0 against 23.4 at 1/32, 13.9 against 18.0 at 1/64.  One rule for every
// way in (reads, k-mer arrays, super-k-mer records, the chunked host leg).
bool partitioned_batch_wanted(const kmu_counter *c, uint64_t n) { return n * (c->empty ? 64u : 10u) >= c->nslots && n >= (1u << 16); }

// the canonical k-mers of device-resident reads into THIS table: the streaming build for big batches of unpacked reads,
// direct insertion otherwise (total_bases: extent of the flat stream, 0 for packed input)
int local_add(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t *d_err) {
    kmu_ctx *ctx = c->ctx;
    KMU_TRY(table_alloc_for(c, &ds, total_bases, d_err));
    bool partitioned = false;
    const char *force = getenv("KMU_COUNT_PATH"); // "direct" / "partitioned": tests
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

// Length of the flat base stream the (unpacked) sequences occupy.  The sequences may be a range of a larger read set
// (offsets[0] > 0, e.g. `offsets + first` of device-resident reads): host input was staged re-based to 0; for device input
// the base pointer is moved forward by whole wave-steps and a re-based copy of the offsets is used, so that neither the
// scratch sizes nor the walk depend on what lies before the range (the kernels skip the few bases left before offsets[0]).
int flat_stream_extent(kmu_ctx *ctx, const uint64_t *host_offsets, uint32_t n_seq, int mem, DevSeqs &ds, uint64_t *total_out) {
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

int add_entries(kmu_counter *c, const uint64_t *kmers, const uint32_t *counts, uint64_t n, int mem) {
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
        bool partitioned = !counts && partitioned_batch_wanted(c, n);
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
int add_superkmers(kmu_counter *c, const void *recs, uint64_t n_rec, uint64_t n_kmers) {
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (n_rec == 0) return KMU_OK;
    uint32_t *d_err;
    KMU_TRY(get_err_word(ctx, &d_err));
    KMU_TRY(table_alloc_for(c, nullptr, 0, d_err, 0.0, std::max<uint64_t>(c->p.capacity_hint, n_kmers)));
    const char *force = getenv("KMU_COUNT_PATH");
    bool partitioned = partitioned_batch_wanted(c, n_kmers);
    if (force && !strcmp(force, "direct")) partitioned = false;
    if (force && !strcmp(force, "partitioned")) partitioned = true;
    PartPlan pl;
    bool overflowed = false;
    if (partitioned && part_plan_for(c, &pl) && pl.b1 && seg_partition_wanted(n_kmers)) {
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

// the largest count a slot of this table holds: what a region build may leave (quotient slots: 2^w - 1024: every adder stops
// there, the plain adds in flight of a region build stay below the field)
static uint64_t count_ceiling(const kmu_counter *c) {
    if (!c->qw) return 0xFFFFFFFFull;
    return (1ull << c->qw) - (uint64_t) Q_MARGIN;
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

// shared by dump / export / finalize: select into device buffers, then hand over
int select_entries(kmu_counter *c, uint32_t min_count, uint32_t maxc, uint32_t part, uint32_t n_parts,
                   uint64_t *kmers_out, uint32_t *counts_out, uint64_t cap, int mem, bool sort, uint64_t *n_out,
                   bool own_owner /* parts by the counter's own owner function instead of the reference's dispatch */) {
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (c->empty) {
        *n_out = 0;
        return KMU_OK;
    }
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

} // namespace kmu

using namespace kmu;

extern "C" {

int kmu_count_reset(kmu_counter *c) {
    if (!c) return KMU_E_BAD_ARG;
    if (c->pending && c->ctx->comm) (void) comm_wait(c->ctx); // an exchange left open: what it writes ("cnt.recv") is dropped, but not under later work
    c->pending = false;
    c->pend_recv = 0;
    c->pend_kmers = 0;
    c->unmerged = false;
    c->empty = true; // materialised lazily: a partitioned build writes every region itself
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
    if (c->lox) (void) hipFree(c->lox);
    if (c->scalars) (void) hipFree(c->scalars);
    delete c;
}

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
