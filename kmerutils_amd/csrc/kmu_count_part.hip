// kmu_count_part.hip -- the radix-partitioned build of the count table (the throughput path of kmu_count_add_reads /
// kmu_sketch_count): big batches never touch the table with atomics, HBM sees only streams.
//
// The bases are consumed as ONE flat stream of aligned 16-byte words; the canonical k-mers travel as khash(k-mer) (kmu_count_table.h)
// and are sorted by the digits of the table's region map -- level 1 by group (the top b1 hash bits), level 2 by sub-region
// (mulhi32 of the next 32 bits with n2) -- with LDS-staged tile sorts, so that the k-mers of one bin leave as contiguous runs;
// then one workgroup per region builds the region in LDS (ds_cmpst / ds_add) and streams its image out.
//  * the SINGLE-PASS partition (default for big batches): no histogram passes -- every stream has a fixed capacity of
//    mean + 5 sigma, shared by the workgroups of a set through a cursor; an item that finds its stream full goes to a spill list
//    (direct insertion behind the build); only a full spill list sends the batch to
//  * the EXACT levels: a histogram pass gives every unit exact private output ranges, then the scatter (also: one-level tables,
//    the owner grouping of a distributed add with hash owners, the generic array partition of the sketch path).
#include <algorithm>
#include <cmath>
#include <vector>

#include "kmu_count_table.h"
#include "kmu_flat.h"
#include "kmu_stream.h"

namespace kmu {

// level 1, pass 1 (exact route; owner census of a distributed add): per-unit histogram of the level-1 digit (also validates the bases)
__global__ void __launch_bounds__(256) k_part_hist1(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq, int k,
                                                    PartPlan pl, uint32_t *hist1, uint32_t *err, SampleArgs sa) {
    extern __shared__ uint32_t lh[];
    const uint32_t bins1 = plan_bins1(pl);
    const Digit d1 = plan_digit1(pl);
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
                atomicAdd(&lh[digit_of_hash(d1, khash(canon))], 1u);
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

// level 1 scan, step a: one workgroup per bin -> exclusive prefix over the units + bin total
__global__ void __launch_bounds__(256) k_part_scan1a(const uint32_t *hist1, PartPlan pl, uint64_t *offs1, uint64_t *tot1) {
    __shared__ uint64_t part[256];
    const uint32_t bins1 = plan_bins1(pl), b = blockIdx.x, U = pl.units1;
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
    const uint32_t bins1 = plan_bins1(pl);
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

// What a partition item is: IT_HASH = khash(key) (digit = a function of the item), IT_KEY = the key itself (digit from
// khash(key)), IT_OWNER = the key, digit = its owner in a key partition (DispatchableT, kmercount.rs:382-420; the Digit then
// carries kmer_owner's mode in `sh` and the number of parts in `n2`), IT_KEY_TO_HASH: keys in, khash(key) out.
enum { IT_HASH = 0, IT_KEY = 1, IT_OWNER = 2, IT_KEY_TO_HASH = 3 };
template <int IT>
__device__ __forceinline__ uint32_t digit_of(uint64_t item, const Digit &d) {
    if (IT == IT_OWNER) return kmer_owner(item, d.sh, d.n2);
    return digit_of_hash(d, IT == IT_HASH ? item : khash(item));
}

// it[j] == CKEY_EMPTY marks "no k-mer".  All 1024 threads call this together.  (The exact levels, the owner grouping of a
// distributed add, the generic array partition; the single-pass partition has its own form, tile_scatter_seg.)
template <int IT>
__device__ __forceinline__ void tile_scatter(uint64_t (&it)[16], const ScatterLds &l, uint32_t nbins, const Digit &d, uint64_t *out) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    uint32_t br[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        br[j] = 0;
        if (it[j] != CKEY_EMPTY) {
            uint32_t bin = digit_of<IT>(it[j], d);
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
            const uint32_t bin = p < total ? digit_of<IT>(v[u], d) : 0u;
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

// ---- the tile sort of the single-pass partition ------------------------------------------------------------------------
// Same tile, same runs, fewer phases: four LDS barriers per tile instead of seven and one table look-up per item on the way
// out instead of two.  The streams are SHARED by the workgroups of a set (level 1: two sets per XCD; level 2: the units of a
// level-1 bin): a tile's run of a bin is placed by an atomic add on the bin's cursor (cursor[bin]: items handed out so far), so
// the runs of the set's workgroups lie one behind the other in ONE stream per bin and the half-written 128-byte lines at the head
// of a stream are completed by the neighbours within a tile's time instead of waiting in L2 for this workgroup's next tile
// (32 workgroups x 2 048 private streams x 128 bytes = 8 MB of open lines per XCD against 4 MB of L2: the two speeds of level 1
// in round 2).  What the write-out needs is one 32-bit word per bin, grel = run start in the stream - start of the bin inside the
// tile (mod 2^32): an item at tile position p goes to slot grel[bin] + p of its stream.  The rank counters are a separate array
// that the owner zeroes while it scans them, so the ranks of the next tile are taken by the waves that are through with this
// tile's write-out while the others still store (no barrier behind the write-out).
//
// 6-byte leaf items (LEAF6): what the region build needs of an item is the 64 - w bits a slot keeps (q_kept) and it knows the
// rest from where it reads; for tables whose count field has w >= 16 bits level 2 leaves those <= 48 bits in 48-byte blocks of
// eight items (eight u32 low words, then eight u16 high parts: the two stores of an item and of its neighbours in a run land
// next to each other): 26 instead of 35 GB written and read back at the bench size.
__device__ __forceinline__ void leaf6_store(uint64_t *out, uint64_t at, uint64_t v) {
    uint8_t *b = reinterpret_cast<uint8_t *>(out) + (at >> 3) * 48u;
    reinterpret_cast<uint32_t *>(b)[at & 7u] = (uint32_t) v;
    reinterpret_cast<uint16_t *>(b + 32)[at & 7u] = (uint16_t) (v >> 32);
}
__device__ __forceinline__ uint64_t leaf6_load(const uint64_t *items, uint64_t at) {
    const uint8_t *b = reinterpret_cast<const uint8_t *>(items) + (at >> 3) * 48u;
    return ((uint64_t) reinterpret_cast<const uint16_t *>(b + 32)[at & 7u] << 32) | reinterpret_cast<const uint32_t *>(b)[at & 7u];
}
// stream (bin_base + bin) holds `cap` items at out[(bin_base + bin) * cap]; an item beyond it goes to the spill list
struct SegOut {
    uint32_t bin_base, cap;
    uint32_t *ovf;
};
struct SegLds {
    uint64_t *stage;  // TILE_ITEMS
    uint32_t *cnt;    // nbins (+ 2 pad): ranks handed out in this tile
    uint32_t *lstart; // nbins (+ 2 pad): exclusive starts inside the tile
    uint32_t *grel;   // nbins
    uint32_t *wtot;   // 16 wave totals
    uint32_t *lox;    // LEAF6: nbins -- the table's lox[] (kmu_count_table.h)
};
__device__ __forceinline__ SegLds seg_lds(uint8_t *smem, uint32_t nbins) {
    SegLds l;
    l.stage = reinterpret_cast<uint64_t *>(smem);
    l.cnt = reinterpret_cast<uint32_t *>(l.stage + TILE_ITEMS);
    l.lstart = l.cnt + nbins + 2;
    l.grel = l.lstart + nbins + 2;
    l.wtot = l.grel + nbins;
    l.lox = l.wtot + 16;
    return l;
}
static size_t seg_lds_bytes(uint32_t nbins, bool lox = false) { return (size_t) TILE_ITEMS * 8 + ((size_t) nbins * 3 + 4) * 4 + 64 + (lox ? (size_t) nbins * 4 : 0); }
static constexpr uint32_t LEAF6_MAX_BINS = 2024; // (with the lox copy the tile sort of 2 048 bins would need 80 bytes more than a CU's 160 KiB)

// An item that finds its stream full goes to the spill list (k-mers that occur many times -- a genome at coverage c -- make
// a bin's fill vary sqrt(c) times more than the margin of independent k-mers allows for; the list is added to the finished
// table by direct insertion, k_count_add_spill); only a full spill list raises the flag that sends the batch to the exact
// levels.  ovf: [0] flag, [1] items spilled, [2] capacity of the list, [4..5] its address.
__device__ __forceinline__ void seg_spill(uint32_t *ovf, uint64_t item) {
    const uint32_t at = atomicAdd(&ovf[1], 1u);
    if (at < ovf[2]) (*reinterpret_cast<uint64_t *const *>(ovf + 4))[at] = item;
    else ovf[0] = 1u;
}

// items are khash values; nbins even, <= 2048; all 1024 threads call this together; cnt[] zero on the first call.
// MUL: the digit is the sub-region (mulhi32 of the 32 bits from bit d.sh on with d.n2: level 2, d.sh in 21 .. 31), else the group
// (the top 64 - d.sh bits: level 1).  VMWAIT: the caller prefetches the next tile with unconditional loads (see flat_step_fetch).
// ALLV: a wave whose sixteen items per lane are all k-mers (nearly every wave of long reads and of the inner levels) takes its ranks
// (bit 0) and stages its items (bit 1) without the per-item branches: the LDS requests of a lane leave back to back and are waited
// for once, not one `s_waitcnt` per item inside sixteen EXEC regions -- for the callers whose registers have the room: level 1 from
// the bases spills 25 with it and takes 18.7 instead of 12.4 ms; the array levels: 17.1 -> 16.7 ms on the bench's level 2.
template <bool VMWAIT, bool MUL, bool LEAF6, int ALLV>
__device__ __forceinline__ void tile_scatter_seg(uint64_t (&it)[16], const SegLds &l, uint32_t nbins, const Digit &d, uint64_t *out,
                                                 const SegOut &sg, uint32_t *cursor) {
    const uint32_t tid = threadIdx.x, nthreads = SCATTER_THREADS;
    const uint32_t sh = MUL ? (uint32_t) d.sh : (uint32_t) d.sh - 32u;
    auto x_of = [&](uint64_t item) -> uint32_t {
        return MUL ? __builtin_amdgcn_alignbit((uint32_t) (item >> 32), (uint32_t) item, sh) : (uint32_t) (item >> 32) >> sh;
    };
    auto bin_of = [&](uint64_t item) -> uint32_t { return MUL ? __umulhi(x_of(item), d.n2) : x_of(item); };
    uint32_t rk[8]; // ranks (< 16384), two to a register
#pragma unroll
    for (int j = 0; j < 8; j++) rk[j] = 0;
    bool allv_lane = true;
#pragma unroll
    for (int j = 0; j < 16; j++) allv_lane = allv_lane && it[j] != CKEY_EMPTY;
    const bool allv = ALLV && __all(allv_lane);
    constexpr int G = 4; // LDS round trips in flight per lane of the branch-free forms (8: 16 / 31 registers spilled)
    if ((ALLV & 1) && allv) {
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
    const uint32_t b0 = 2u * tid;
    uint32_t c0 = 0, c1 = 0, run0 = 0, run1 = 0;
    if (b0 < nbins) {
        const uint2 c = *reinterpret_cast<const uint2 *>(&l.cnt[b0]);
        c0 = c.x;
        c1 = c.y;
        *reinterpret_cast<uint2 *>(&l.cnt[b0]) = make_uint2(0u, 0u);
        run0 = atomicAdd(&cursor[b0], c0); // (the answers are looked at behind the staging)
        run1 = atomicAdd(&cursor[b0 + 1], c1);
    }
    const uint32_t incl = wave_incl_scan_u32(c0 + c1);
    if (lane_id() == 63) l.wtot[tid >> 6] = incl;
    lds_barrier();
    uint32_t wpre = 0, total = 0; // total: the k-mers of the tile
    {
        const uint4 *w4 = reinterpret_cast<const uint4 *>(l.wtot);
        const uint32_t wave = tid >> 6;
#pragma unroll
        for (int q = 0; q < SCATTER_THREADS / 256; q++) {
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
    }
    lds_barrier();
    if ((ALLV & 2) && allv) {
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
    if (b0 < nbins) {
        const uint2 ls = *reinterpret_cast<const uint2 *>(&l.lstart[b0]);
        *reinterpret_cast<uint2 *>(&l.grel[b0]) = make_uint2(run0 - ls.x, run1 - ls.y);
    }
    lds_barrier();
    if (VMWAIT) vm_wait_all(); // the next tile's requests (in flight since before the ranks) and the last tile's stores: nothing younger
    const uint32_t bb = sg.bin_base, cap = sg.cap;
    const uint64_t lowmask = (1ull << sh) - 1ull; // (LEAF6: the hash bits below x)
    for (uint32_t p0 = 0; p0 < total; p0 += 8u * nthreads) {
        uint64_t v[8];
        uint32_t rel[8], bin[8], lx[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = p0 + (uint32_t) u * nthreads + tid;
            v[u] = l.stage[p < total ? p : 0u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = p0 + (uint32_t) u * nthreads + tid;
            bin[u] = bin_of(v[u]);
            rel[u] = l.grel[bin[u]] + p;
            lx[u] = LEAF6 ? l.lox[bin[u]] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = p0 + (uint32_t) u * nthreads + tid;
            if (p < total) {
                const uint64_t at = (uint64_t) (bb + bin[u]) * cap + rel[u];
                if (rel[u] >= cap) seg_spill(sg.ovf, v[u]);
                else if (LEAF6) leaf6_store(out, at, ((uint64_t) (x_of(v[u]) - lx[u]) << sh) | (v[u] & lowmask));
                else out[at] = v[u];
            }
        }
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
// halo word of lane & 1) and the lane's 16 "no k-mer" bits (flat_novalid, kmu_smer.hpp), and nothing looks at them until
// flat_step_words turns them into code words one tile later -- the requests are UNCONDITIONAL loads from clamped addresses
// (needs total >= 16), so that their number in flight is a constant for the compiler's s_waitcnt placement (a load under a
// branch makes it wait for everything at the first use of anything).  All vector-memory waits of the scatter loops are the
// explicit vmcnt(0) of vm_wait_all(): once before the loop, once per tile just before the write-out, when the requests of the
// next tile have had the whole tile sort to arrive and the stores of the last tile are long gone.
struct FlatRaw {
    uint4 c0, cx;
    uint32_t nv;
};
__device__ __forceinline__ void flat_step_fetch(const uint8_t *bases, uint64_t total, uint64_t st, FlatRaw &r, const uint16_t *novalid, uint64_t last_step) {
    r.c0 = make_uint4(0u, 0u, 0u, 0u);
    r.cx = r.c0;
    r.nv = novalid[(st < last_step ? st : last_step) * 64 + (uint64_t) lane_id()];
    if (total < 16) return; // (wave-uniform; flat_step_words then reads the ragged chunk itself)
    const uint64_t lastc = (total - 16) & ~15ull;
    const uint64_t a0 = (st * 64 + (uint64_t) lane_id()) * 16, ax = (st * 64 + 64 + (uint64_t) (lane_id() & 1)) * 16;
    r.c0 = *reinterpret_cast<const uint4 *>(bases + (a0 < lastc ? a0 : lastc));
    r.cx = *reinterpret_cast<const uint4 *>(bases + (ax < lastc ? ax : lastc));
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

// up to 16 canonical k-mers of this lane for wave step `st` (CKEY_EMPTY where a k-mer would straddle a read end); the exact levels
__device__ __forceinline__ void flat_step_items(const uint64_t *offsets, uint32_t n_seq, uint64_t total, uint64_t start, int k,
                                                uint64_t st, bool active, uint32_t w0, uint32_t ex, uint32_t &r_hint, uint64_t (&it)[16]) {
#pragma unroll
    for (int j = 0; j < 16; j++) it[j] = CKEY_EMPTY;
    if (!active) return; // wave-uniform
    const int lane = lane_id();
    const uint64_t widx = st * 64 + lane;
    uint32_t e0 = bcast_u32(ex, 0), e1 = bcast_u32(ex, 1);
    uint32_t w1 = shfl_down_u32(w0, 1), w2 = shfl_down_u32(w0, 2);
    if (lane == 63) { w1 = e0; w2 = e1; }
    if (lane == 62) { w2 = e0; }
    const uint64_t g0 = widx * 16;
    const bool in = g0 < total && g0 + 16 > start;
    uint64_t rend = 0;
    uint32_t r = wave_find_read_from(offsets, n_seq, st * 1024 < total ? st * 1024 : total - 1, r_hint);
    r_hint = r;
    if (in) {
        rend = offsets[r + 1];
        while (g0 >= rend && r + 1 < n_seq) { r++; rend = offsets[r + 1]; } // the read of this lane's first base
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
            while (g >= rend && r + 1 < n_seq) { r++; rend = offsets[r + 1]; }
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
            for (int j = 0; j < 16; j++) it[j] = step_canonical(sw, j);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++)
            if ((V >> j) & 1u) it[j] = step_canonical(sw, j);
    }
}

// level 1, pass 2 of the exact route (private ranges per unit from the histogram), and the owner grouping of a distributed add
__global__ void __launch_bounds__(1024) k_part_scatter1_exact(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq,
                                                              int k, PartPlan pl, const uint64_t *offs1,
                                                              const uint64_t *binstart1, uint64_t *out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t bins1 = plan_bins1(pl);
    ScatterLds l = scatter_lds(smem, bins1);
    for (uint32_t b = threadIdx.x; b < bins1; b += blockDim.x) {
        l.gbase[b] = binstart1[b] + offs1[(uint64_t) blockIdx.x * bins1 + b];
        l.lstart[b] = 0;
    }
    if (threadIdx.x == 0) l.lstart[bins1] = 0;
    lds_barrier();
    const uint64_t total = offsets[n_seq], start = offsets[0];
    const uint64_t nsteps = ((total + 15) / 16 + 63) / 64;
    const uint64_t s0 = (uint64_t) blockIdx.x * pl.steps_per_unit;
    const uint64_t s1 = s0 + pl.steps_per_unit < nsteps ? s0 + pl.steps_per_unit : nsteps;
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t r_hint = 0xFFFFFFFFu, w0, ex;
    const Digit d1 = plan_digit1(pl);
    for (uint64_t t0 = s0; t0 < s1; t0 += nwaves) {
        uint64_t it[16];
        flat_step_load(bases, total, t0 + wave, t0 + wave < s1, w0, ex);
        flat_step_items(offsets, n_seq, total, start, k, t0 + wave, t0 + wave < s1, w0, ex, r_hint, it);
        if (pl.owner_parts) tile_scatter<IT_OWNER>(it, l, bins1, Digit{pl.owner_w32, pl.owner_parts}, out);
        else { // from here on the k-mers travel as their table hash
#pragma unroll
            for (int j = 0; j < 16; j++) it[j] = khash(it[j]); // (khash keeps the "no k-mer" mark)
            tile_scatter<IT_HASH>(it, l, bins1, d1, out);
        }
    }
}

// level 1 of the single-pass partition: flat base stream -> canonical k-mer -> khash -> tile sort by group -> shared streams
// [set][bin][cap] through the cursors `state`[set][bin] (set = blockIdx.x % sets: the dispatcher deals the workgroups out to the
// XCDs round robin).  The kernel validates the bases (no histogram pass ran); k_seg_tails marks the tails behind the last launch.
// Rounds (kmu_sketch_count under an upload): the same units take a slice of every round's wave steps [step_base, step_end).
struct SegPlan1 {
    uint64_t cap, step_base, step_end;
    uint32_t *ovf, *err, *state;
    uint32_t sets;
    const uint16_t *novalid; // flat_novalid's bits of all wave steps of the stream
};
__global__ void __launch_bounds__(1024) k_part_scatter1(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seq,
                                                        int k, PartPlan pl, uint64_t *out, SegPlan1 seg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t bins1 = plan_bins1(pl);
    SegLds ls = seg_lds(smem, bins1);
    const uint32_t set = blockIdx.x % seg.sets;
    const SegOut sg{set * bins1, (uint32_t) seg.cap, seg.ovf};
    uint32_t *cursor = seg.state + (size_t) set * bins1;
    for (uint32_t b = threadIdx.x; b < bins1 + 2; b += blockDim.x) ls.cnt[b] = 0;
    lds_barrier();
    const uint64_t total = offsets[n_seq];
    const uint64_t nsteps_all = ((total + 15) / 16 + 63) / 64;
    const uint64_t nsteps = seg.step_end ? seg.step_end : nsteps_all;
    const uint64_t s0 = seg.step_base + (uint64_t) blockIdx.x * pl.steps_per_unit;
    const uint64_t s1 = s0 + pl.steps_per_unit < nsteps ? s0 + pl.steps_per_unit : nsteps;
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t w0, ex, bad = 0;
    FlatRaw raw;
    const uint64_t last_step = nsteps_all ? nsteps_all - 1 : 0;
    flat_step_fetch(bases, total, s0 + wave, raw, seg.novalid, last_step);
    vm_wait_all();
    const Digit d1 = plan_digit1(pl);
    for (uint64_t t0 = s0; t0 < s1; t0 += nwaves) {
        uint64_t it[16];
        flat_step_words(bases, total, t0 + wave, t0 + wave < s1, raw, w0, ex, bad);
        flat_step_items_nv(k, t0 + wave < s1, w0, ex, raw.nv, it);
        // the next step's chunks and its "no k-mer" bits are requested now; they arrive under the tile sort, which waits for them
        // before its write-out
        flat_step_fetch(bases, total, t0 + nwaves + wave, raw, seg.novalid, last_step);
#pragma unroll
        for (int j = 0; j < 16; j++) it[j] = khash(it[j]); // from here on the k-mers travel as their table hash (khash keeps the "no k-mer" mark)
        tile_scatter_seg<true, false, false, 0>(it, ls, bins1, d1, out, sg, cursor);
    }
    if (bad) atomicOr(seg.err, DERR_NON_ACGT);
}

// ---- generic radix partition of a u64 array (level 2 of the read path; both levels of the array path) ------------
// The input is a set of `nparts` consecutive partitions (bounds[nparts + 1]); every partition is cut into `chunks`
// units; a unit scatters its slice by the digit `d` into `bins` sub-partitions.
struct ArrPlan {
    Digit d;
    uint32_t bins;
    uint32_t nparts;
    uint32_t chunks;
    // single-pass form: != 0: the input partition p is not contiguous but the p-th stream (seg_cap items) of each of seg_units
    // blocks of seg_bins streams -- what the single-pass level 1 leaves, one block per set; bounds is not read
    uint32_t seg_units, seg_cap, seg_bins;
    // single-pass form, level 1 of an array: > 1 = that many sets of shared output streams, a unit writes set blockIdx.x % out_sets
    // ([set][bin][cap], cursors in the same order); 0 / 1: one set per input partition (level 2: the leaves)
    uint32_t out_sets;
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
        atomicAdd(&lh[digit_of<IT>(in[i], pl.d)], 1u);
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

// the exact route: unit (partition, chunk) writes bin b into its private range from the histogram
template <int IT>
__global__ void __launch_bounds__(SCATTER_THREADS) k_arr_scatter_exact(const uint64_t *in, const uint64_t *bounds, ArrPlan pl,
                                                                       const uint64_t *offs_rel, const uint64_t *outbounds, uint64_t *out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t TILE = 16u * SCATTER_THREADS;
    ScatterLds l = scatter_lds(smem, pl.bins);
    for (uint32_t b = threadIdx.x; b < pl.bins; b += blockDim.x) {
        l.gbase[b] = outbounds[(uint64_t) (blockIdx.x / pl.chunks) * pl.bins + b] + offs_rel[(uint64_t) blockIdx.x * pl.bins + b];
        l.lstart[b] = 0;
    }
    if (threadIdx.x == 0) l.lstart[pl.bins] = 0;
    lds_barrier();
    uint64_t i0, i1;
    arr_unit_range(bounds, pl, blockIdx.x, &i0, &i1);
    uint64_t nxt[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint64_t i = i0 + (uint64_t) j * blockDim.x + threadIdx.x;
        nxt[j] = i < i1 ? in[i] : CKEY_EMPTY;
    }
    for (uint64_t t0 = i0; t0 < i1; t0 += TILE) {
        uint64_t it[16];
#pragma unroll
        for (int j = 0; j < 16; j++) it[j] = nxt[j];
        // the next tile is requested before this one is sorted: its HBM latency hides under the LDS work
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint64_t i = t0 + TILE + (uint64_t) j * blockDim.x + threadIdx.x;
            nxt[j] = i < i1 ? in[i] : CKEY_EMPTY;
        }
        if (IT == IT_KEY_TO_HASH) { // from here on the k-mers travel as their table hash (no further evaluations)
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (it[j] != CKEY_EMPTY) it[j] = khash(it[j]);
        }
        tile_scatter<IT == IT_KEY_TO_HASH ? IT_HASH : IT>(it, l, pl.bins, pl.d, out);
    }
}

// The single-pass form: no histogram ran; "no k-mer" marks in the input (the tails of the previous level's streams) are skipped
// like everywhere else.
//  IT_HASH (level 2 of the read path and of the array path): the `chunks` units of an input partition (a level-1 bin) write ONE
//   set of leaves, a tile's run of a leaf placed by an atomic add on the leaf's cursor (leafcnt[leaf], zero before the launch; it
//   ends as the leaf's fill -- or more, where items went to the spill list: the build clamps it).  The units of a partition are the
//   workgroups 8 apart in the grid: the dispatcher deals workgroups out to the 8 XCDs round robin, so they run at the same time on
//   the same XCD and its L2 sees their runs of a leaf side by side.  The input is requested a tile ahead, whole, by unconditional
//   loads (positions beyond the partition are mapped to its last block and not looked at), the waits are explicit.
//  IT_KEY_TO_HASH (level 1 of an array of canonical k-mers): keys in, khash out, pl.out_sets sets of shared streams.
template <int IT, bool LEAF6>
__global__ void __launch_bounds__(SCATTER_THREADS) k_arr_scatter_seg(const uint64_t *in, const uint64_t *bounds, ArrPlan pl, uint64_t *out,
                                                                     uint64_t seg_cap, uint32_t *seg_ovf, uint32_t *leafcnt, const uint32_t *lox) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t TILE = 16u * SCATTER_THREADS, THREADS = SCATTER_THREADS;
    constexpr bool L2 = IT == IT_HASH;
    SegLds ls = seg_lds(smem, pl.bins);
    uint64_t sp = blockIdx.x / pl.chunks;
    uint32_t my_chunk = blockIdx.x % pl.chunks;
    if (pl.seg_units && (pl.nparts & 7u) == 0u) {
        sp = 8u * (blockIdx.x / (8u * pl.chunks)) + (blockIdx.x & 7u);
        my_chunk = (blockIdx.x >> 3) % pl.chunks;
    }
    const uint32_t nsets = pl.out_sets > 1u ? pl.out_sets : 1u;
    const uint64_t block = sp * nsets + (nsets > 1u ? blockIdx.x % nsets : 0u);
    const SegOut sg{(uint32_t) (block * pl.bins), (uint32_t) seg_cap, seg_ovf};
    uint32_t *cursor = leafcnt + block * pl.bins;
    for (uint32_t b = threadIdx.x; b < pl.bins + 2; b += blockDim.x) ls.cnt[b] = 0;
    if (LEAF6)
        for (uint32_t b = threadIdx.x; b < pl.bins; b += blockDim.x) ls.lox[b] = lox[b];
    lds_barrier();
    uint64_t i0, i1;
    if (pl.seg_units) { // (positions in the partition's streams, one after the other): this unit's slice, from a multiple of 16 positions on
        i1 = (uint64_t) pl.seg_units * pl.seg_cap;
        const uint64_t per = ((i1 + pl.chunks - 1) / pl.chunks + 15) & ~(uint64_t) 15;
        i0 = (uint64_t) my_chunk * per < i1 ? (uint64_t) my_chunk * per : i1;
        i1 = i0 + per < i1 ? i0 + per : i1;
    } else arr_unit_range(bounds, pl, blockIdx.x, &i0, &i1);
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    uint64_t nxt[16];
    // L2: 16 bytes per lane and request -- item 2 j2 + e of a thread is element j2 * 2 THREADS + 2 tid + e of the tile (the
    // unit starts on a multiple of 16 items: stream capacities are multiples of 16)
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint64_t i = i0 + (uint64_t) j * blockDim.x + threadIdx.x;
        if (!L2) nxt[j] = i < i1 ? in[i] : CKEY_EMPTY;
    }
    // the 16 items of this thread of the tile that starts at position t of the partition
    const uint64_t seg_stride = (uint64_t) pl.seg_bins * pl.seg_cap, seg_base = (uint64_t) sp * pl.seg_cap;
    auto tile_request = [&](uint64_t t) {
        // position -> (block, offset); a pair of items never straddles streams (their sizes are multiples of 16)
        uint32_t i = (uint32_t) t + 2u * threadIdx.x;
        uint32_t u = i / pl.seg_cap, o = i - u * pl.seg_cap;
#pragma unroll
        for (int j2 = 0; j2 < 8; j2++) {
            // a pair beyond the unit's slice is not looked at: its lanes ask for the partition's first pair, one line that the
            // vector cache holds (requested as whole tiles a unit of 8.3 tiles fetched 10: 42.7 GB for the bench's 35.6)
            const bool mine = t + (uint64_t) j2 * (2u * THREADS) + 2u * threadIdx.x < i1;
            const u64x2 q = *reinterpret_cast<const u64x2 *>(in + (mine ? (uint64_t) u * seg_stride + seg_base + o : seg_base));
            nxt[2 * j2] = q.x;
            nxt[2 * j2 + 1] = q.y;
            o += 2u * THREADS;
            if (pl.seg_cap >= 2u * THREADS) { if (o >= pl.seg_cap) { o -= pl.seg_cap; u++; } }
            else { const uint32_t dd = o / pl.seg_cap; u += dd; o -= dd * pl.seg_cap; }
        }
    };
    if (L2) {
        if (i0 < i1) tile_request(i0);
        else {
#pragma unroll
            for (int j = 0; j < 16; j++) nxt[j] = CKEY_EMPTY;
        }
        vm_wait_all();
    }
    for (uint64_t t0 = i0; t0 < i1; t0 += TILE) {
        uint64_t it[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (L2) it[j] = t0 + (uint64_t) (j >> 1) * (2u * THREADS) + 2u * threadIdx.x + (j & 1) < i1 ? nxt[j] : CKEY_EMPTY;
            else it[j] = nxt[j];
        }
        // the next tile is requested before this one is sorted: its HBM latency hides under the LDS work
        if (L2) tile_request(t0 + TILE);
        else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint64_t i = t0 + TILE + (uint64_t) j * blockDim.x + threadIdx.x;
                nxt[j] = i < i1 ? in[i] : CKEY_EMPTY;
            }
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (it[j] != CKEY_EMPTY) it[j] = khash(it[j]); // from here on the k-mers travel as their table hash
        }
        tile_scatter_seg<L2, L2, LEAF6, 3>(it, ls, pl.bins, pl.d, out, sg, cursor);
    }
}

// Level 1 of the receiver of a super-k-mer exchange (kmu_smer.h): the input is an array of 12-byte records, a thread takes one
// record per tile and expands it into its <= 16 canonical k-mers with the window arithmetic of the read path (a record IS the
// lane's three code words); from there on the tile sort of the single-pass partition, shared streams and cursors as in
// k_arr_scatter_seg<IT_KEY_TO_HASH>.  Unit u of `chunks` takes records [n u / chunks, n (u + 1) / chunks).
__global__ void __launch_bounds__(SCATTER_THREADS) k_smer_scatter1(const uint32_t *recs, uint64_t n_rec, int k, ArrPlan pl, uint64_t *out,
                                                                  uint64_t seg_cap, uint32_t *seg_ovf, uint32_t *cursors) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SegLds ls = seg_lds(smem, pl.bins);
    const uint32_t nsets = pl.out_sets > 1u ? pl.out_sets : 1u;
    const uint32_t block = nsets > 1u ? blockIdx.x % nsets : 0u;
    const SegOut sg{block * pl.bins, (uint32_t) seg_cap, seg_ovf};
    uint32_t *cursor = cursors + (size_t) block * pl.bins;
    for (uint32_t b = threadIdx.x; b < pl.bins + 2; b += blockDim.x) ls.cnt[b] = 0;
    lds_barrier();
    const uint64_t i0 = n_rec * blockIdx.x / pl.chunks, i1 = n_rec * (blockIdx.x + 1) / pl.chunks;
    uint32_t nx0 = 0, nx1 = 0, nx2 = 0;
    bool nxv = false;
    auto fetch = [&](uint64_t i) {
        nxv = i < i1;
        if (nxv) { nx0 = recs[i * 3]; nx1 = recs[i * 3 + 1]; nx2 = recs[i * 3 + 2]; }
    };
    fetch(i0 + threadIdx.x);
    for (uint64_t t0 = i0; t0 < i1; t0 += SCATTER_THREADS) {
        const uint32_t w0 = nx0, w1 = nx1, w2 = nx2, L = nxv ? (nx2 & 15u) + 1u : 0u;
        fetch(t0 + SCATTER_THREADS + threadIdx.x); // the next tile's record arrives under this tile's sort
        uint64_t it[16];
        const StepWin sw = step_win(w0, w1, w2 & ~15u, k); // (the low four bits of a record's last word: its k-mer count)
#pragma unroll
        for (int j = 0; j < 16; j++)
            it[j] = (uint32_t) j < L ? khash(step_canonical(sw, j)) : CKEY_EMPTY; // kmer.reverse_complement().min(kmer), kmercount.rs:938
        tile_scatter_seg<false, false, false, 0>(it, ls, pl.bins, pl.d, out, sg, cursor);
    }
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

// "no k-mer" marks from the fill of every (set, bin) stream of level 1 to its capacity (level 2 reads whole streams)
// (a few workgroups per CU, each over many streams, 16 bytes per lane: one workgroup per stream -- 32 768 launches of 15 KB at the
//  bench size -- took 0.7 ms for 0.5 GB)
__global__ void __launch_bounds__(256) k_seg_tails(const uint32_t *cursor, uint32_t n_streams, uint32_t cap, uint64_t *out) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    for (uint32_t s = blockIdx.x; s < n_streams; s += gridDim.x) {
        uint32_t n = cursor[s] < cap ? cursor[s] : cap;
        uint64_t *o = out + (uint64_t) s * cap;
        if ((n & 1u) && n < cap) { // (cap is even: pairs from an even position on)
            if (threadIdx.x == 0) o[n] = CKEY_EMPTY;
            n++;
        }
        for (uint32_t i = n + 2u * threadIdx.x; i < cap; i += 2u * blockDim.x) *reinterpret_cast<u64x2 *>(o + i) = u64x2{CKEY_EMPTY, CKEY_EMPTY};
    }
}

// the spill list of a single-pass partition (khash values) into the finished table, by direct insertion
__global__ void __launch_bounds__(256) k_count_add_spill(const uint64_t *items, const uint32_t *ovf, CountTable t, uint32_t *err) {
    const uint32_t n = ovf[1] < ovf[2] ? ovf[1] : ovf[2];
    bool full = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (!count_insert_h(t, t.w ? 0ull : khash_inv(items[i]), items[i], 1u)) full = true;
    if (full) atomicOr(err, DERR_TABLE_FULL);
}

// out[i] = i * stride
__global__ void __launch_bounds__(256) k_fill_linear(uint64_t *out, uint64_t n, uint64_t stride) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = i * stride;
}

// ---- the region build: one workgroup per region --------------------------------------------------------------------------
// The region lives in LDS while its k-mers are inserted, then it leaves for HBM.  in_mode: 0 = the table holds nothing yet (no
// region is read), 1 = the slab is read first.  The first BUILD_PRE items of every thread are requested before the region is
// initialised, so their HBM latency hides under the LDS fill; the workgroups of a CU overlap each other's phases.
static constexpr int BUILD_THREADS = 1 << (REGION_BITS_MAX - 3); // 512 threads for regions of 4096 slots
static constexpr int BUILD_PRE = 6;
// batched form of the quotient build: entries of the per-wave pool of items that two probes did not place (8 bytes each, behind the
// region in LDS: 32 + 6 KiB per workgroup, four workgroups per CU as before)
static constexpr uint32_t BUILD_POOL = 96, BUILD_PASSES = 3; // batched probes of an item before it goes to the pool (1 / 2 / 3: 21.1 / 18.8 / 18.4 ms)

// where the items of region r lie: [leafstart[r], leafstart[r + 1]) (exact route), or a fixed-size leaf with its fill in leafcnt
// (single-pass route: the fill may exceed the capacity where items went to the spill list)
__device__ __forceinline__ void leaf_range(uint32_t r, const uint64_t *leafstart, uint64_t leaf_stride, const uint32_t *leafcnt, uint64_t &i0, uint64_t &i1) {
    if (leaf_stride) {
        i0 = (uint64_t) r * leaf_stride;
        i1 = i0 + (leafcnt[r] < leaf_stride ? (uint64_t) leafcnt[r] : leaf_stride);
    } else {
        i0 = leafstart ? leafstart[r] : 0;
        i1 = leafstart ? leafstart[r + 1] : 0;
    }
}

// Quotient slots: the region is 4 096 8-byte words in LDS (32 KiB: four workgroups of 512 threads per CU), a first sighting is
// one ds_cmpst_rtn_b64, a repeat one more ds_add_u64 (guarded: the count field stops short of its width), and the LDS image leaves as
// it is.  LEAF6: the leaves hold the <= 48 bits a slot keeps of an item (tile_scatter_seg) instead of its hash.
// The items of a workgroup's NEXT region are requested before this region is built, and the bounds of the one after that with
// them (round 5): a leaf's items come from HBM at the latency of a memory system that the builds themselves keep busy, and a
// workgroup that asks at the top of its region (first for the leaf's fill, then for the items) waits for them behind the LDS fill
// with only the three other workgroups of its CU to cover for it.
template <int IT, bool LEAF6>
__global__ void __launch_bounds__(BUILD_THREADS, 8) k_part_build_q(const uint64_t *__restrict__ items, const uint64_t *__restrict__ leafstart,
                                                                uint32_t n_regions, CountTable t, int in_mode, uint32_t *err,
                                                                uint64_t leaf_stride, const uint32_t *__restrict__ leafcnt) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t R = t.rmask + 1;
    uint64_t *lk = reinterpret_cast<uint64_t *>(smem);
    uint4 *lk4 = reinterpret_cast<uint4 *>(lk);
    const uint32_t tid = threadIdx.x;
    uint64_t *pool = lk + R + (uint32_t) __builtin_amdgcn_readfirstlane((int) (tid >> 6)) * BUILD_POOL; // this wave's
    uint32_t full = 0;
    const int w = t.w, xs = 32 - t.b1, os = 32 - t.rbits;
    const uint64_t cmask = q_cmask(w), lowmask = (1ull << xs) - 1ull;
    const uint64_t add_limit = q_limit(w);
    auto item_at = [&](uint64_t i) -> uint64_t { return LEAF6 ? leaf6_load(items, i) : items[i]; };
    // where the items of region rr lie (nothing for a region beyond the table), and the first BUILD_PRE of them of this thread
    auto range = [&](uint32_t rr, uint64_t &a0, uint64_t &a1) {
        a0 = a1 = 0;
        if (rr < n_regions) leaf_range(rr, leafstart, leaf_stride, leafcnt, a0, a1);
    };
    // the requested items wait a region long in registers: as nine words where they are 48-bit leaves (twelve otherwise)
    struct Ahead {
        uint32_t lo[BUILD_PRE], hi[LEAF6 ? BUILD_PRE / 2 : BUILD_PRE];
    };
    auto request = [&](uint64_t a0, uint64_t a1, Ahead &it) {
#pragma unroll
        for (int q = 0; q < BUILD_PRE; q++) {
            const uint64_t i = a0 + (uint64_t) q * BUILD_THREADS + tid;
            if (LEAF6) {
                const uint8_t *b = reinterpret_cast<const uint8_t *>(items) + (i >> 3) * 48u;
                it.lo[q] = i < a1 ? reinterpret_cast<const uint32_t *>(b)[i & 7u] : ~0u;
                const uint32_t h = i < a1 ? (uint32_t) reinterpret_cast<const uint16_t *>(b + 32)[i & 7u] : 0xFFFFu;
                it.hi[q >> 1] = (q & 1) ? it.hi[q >> 1] | (h << 16) : h;
            } else {
                const uint64_t v = i < a1 ? items[i] : CKEY_EMPTY;
                it.lo[q] = (uint32_t) v;
                it.hi[q] = (uint32_t) (v >> 32);
            }
        }
    };
    // (a 48-bit leaf item is never all ones: the "no item" mark of a lane beyond the leaf's fill is 2^48 - 1)
    auto unpack = [&](const Ahead &it, int q) -> uint64_t {
        if (!LEAF6) return ((uint64_t) it.hi[q] << 32) | it.lo[q];
        const uint64_t v = ((uint64_t) ((it.hi[q >> 1] >> (16 * (q & 1))) & 0xFFFFu) << 32) | it.lo[q];
        return v == 0xFFFFFFFFFFFFull ? CKEY_EMPTY : v;
    };
    Ahead nxt_it;
    uint64_t n0 = 0, n1 = 0, m0 = 0, m1 = 0;
    range(blockIdx.x, n0, n1);
    request(n0, n1, nxt_it);
    range(blockIdx.x + gridDim.x, m0, m1);
    if (in_mode != 1) {
        for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) lk4[s] = make_uint4(~0u, ~0u, ~0u, ~0u);
        lds_barrier();
    }
    for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
        uint64_t pre_it[BUILD_PRE];
        const uint64_t i0 = n0, i1 = n1;
#pragma unroll
        for (int q = 0; q < BUILD_PRE; q++) pre_it[q] = unpack(nxt_it, q);
        n0 = m0;
        n1 = m1;
        request(n0, n1, nxt_it);                 // region r + grid: its bounds have been here since the last turn
        range(r + 2u * gridDim.x, m0, m1);       // region r + 2 grid: looked at in the next turn
        const uint32_t lox_s = t.lox[r % t.n2]; // (workgroup-uniform: the sub-region of this region)
        uint4 *gk4 = reinterpret_cast<uint4 *>(t.keys + (uint64_t) r * R);
        if (in_mode == 1) { // (an empty table: the region in LDS is empty already -- the copy-out of the last one left it so)
            for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) lk4[s] = gk4[s];
            lds_barrier();
        }
        // an item -> the slot word it would claim (count 0) and its home slot
        auto locate = [&](uint64_t item, uint64_t &hw, uint32_t &off) {
            if (LEAF6) {
                const uint32_t x = lox_s + (uint32_t) (item >> xs);
                off = (x * t.n2) >> os;
                hw = item << w;
            } else {
                const uint64_t h = IT == IT_HASH ? item : khash(item);
                const uint32_t x = (uint32_t) (h >> xs);
                off = (x * t.n2) >> os;
                hw = (((uint64_t) (x - lox_s) << xs) | (h & lowmask)) << w;
            }
        };
        // one probe of `item` at slot `off`: true = the item is in (claimed a free slot, or met its own key)
        auto probe = [&](uint64_t hw, uint32_t off) -> bool {
            const unsigned long long old = atomicCAS((unsigned long long *) &lk[off], (unsigned long long) CKEY_EMPTY, (unsigned long long) (hw | 1ull));
            if (old == CKEY_EMPTY) return true;
            if (!q_same(old, hw, w)) return false;
            // the adds in flight behind a count seen below the limit are fewer than the margin: no carry into the key bits
            if ((old & cmask) < add_limit) atomicAdd((unsigned long long *) &lk[off], 1ull);
            return true;
        };
        // the rest of an item's probe sequence from its n-th probe at `off` on
        auto walk = [&](uint64_t hw, uint32_t off, uint32_t n) {
            do {
                off = (off + ++n) & t.rmask;
                if (n >= R) { full = 1; break; }
            } while (!probe(hw, off));
        };
        {
            // Item by item (rounds 2-4: every lane walking through its six items at its own pace) a wave repeats "probe, wait for the
            // answer, branch" until the unluckiest of its lanes is through -- ~31 trips of ~55 instructions where the average item
            // needs 1.6 probes; 21.3 ms for the bench's table where this form takes 20.3 (9.4 -> 7.2e9 vector, 8.2 -> 5.6e9 scalar
            // wave-instructions).  Here the first
            // probes of a thread's six items leave back to back without a branch (a lane without an item compares against 0, which no
            // slot holds, so nothing is written) and are waited for once; so do the second probes (the next slot) of the items that
            // failed; what is left -- one item in seven -- is pooled per WAVE in LDS and the lanes take the pool's entries, one each:
            // the walk of the remaining probe sequences is as long as the longest of them, not as the unluckiest lane's sum.
            // Two items at a time: a thread that has seen counts below the ceiling has at most two adds in flight behind them, the 512
            // threads 1024 = Q_MARGIN -- a field stops at 2^w - 1 at the latest, as with the item-by-item loop.
            constexpr int H = 2;
            static_assert(BUILD_THREADS * H <= (int) Q_MARGIN && BUILD_PRE % H == 0, "adds in flight against the margin of a count field");
            uint32_t n_pool = 0; // (wave-uniform)
#pragma unroll
            for (int h = 0; h < BUILD_PRE / H; h++) {
                uint64_t hw[H];
                uint32_t off[H];
                bool todo[H];
#pragma unroll
                for (int q = 0; q < H; q++) {
                    locate(pre_it[H * h + q], hw[q], off[q]);
                    todo[q] = pre_it[H * h + q] != CKEY_EMPTY;
                }
#pragma unroll
                for (int pass = 0; pass < BUILD_PASSES; pass++) {
                    unsigned long long old[H];
#pragma unroll
                    for (int q = 0; q < H; q++) {
                        if (pass == 0) // (nearly every lane has an item)
                            old[q] = atomicCAS((unsigned long long *) &lk[off[q]], todo[q] ? (unsigned long long) CKEY_EMPTY : 0ull, (unsigned long long) (hw[q] | 1ull));
                        else {         // (one lane in three)
                            off[q] = (off[q] + (uint32_t) pass) & t.rmask;
                            old[q] = 0ull;
                            if (todo[q]) old[q] = atomicCAS((unsigned long long *) &lk[off[q]], (unsigned long long) CKEY_EMPTY, (unsigned long long) (hw[q] | 1ull));
                        }
                    }
#pragma unroll
                    for (int q = 0; q < H; q++) {
                        const bool claimed = old[q] == CKEY_EMPTY, same = !claimed && q_same(old[q], hw[q], w);
                        if (todo[q] && same && (old[q] & cmask) < add_limit) atomicAdd((unsigned long long *) &lk[off[q]], 1ull);
                        todo[q] = todo[q] && !claimed && !same;
                    }
                }
#pragma unroll
                for (int q = 0; q < H; q++) {
                    const uint64_t m = __ballot(todo[q]);
                    const uint32_t at = n_pool + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
                    if (todo[q]) {
                        if (at < BUILD_POOL) pool[at] = hw[q];
                        else walk(hw[q], off[q], BUILD_PASSES - 1u); // (a wave with more than 96 of 384 items left after the batched probes: a table that is filling up)
                    }
                    n_pool += (uint32_t) __popcll(m);
                }
            }
            n_pool = n_pool < BUILD_POOL ? n_pool : BUILD_POOL;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t e = tid & 63u; e < n_pool; e += 64u) {
                const uint64_t h = pool[e];
                const uint32_t x = lox_s + (uint32_t) ((h >> w) >> xs);
                walk(h, (((x * t.n2) >> os) + (BUILD_PASSES - 1u) * BUILD_PASSES / 2u) & t.rmask, BUILD_PASSES - 1u); // (its last probe: the triangular number)
            }
            __builtin_amdgcn_wave_barrier(); // (the pool is this wave's alone: the next region's entries come behind two workgroup barriers)
        }
        for (uint64_t i = i0 + (uint64_t) BUILD_PRE * BUILD_THREADS + tid; i < i1; i += BUILD_THREADS) { // (leaves beyond 3 072 items)
            const uint64_t item = item_at(i);
            if (item == CKEY_EMPTY) continue;
            uint64_t hw;
            uint32_t off;
            locate(item, hw, off);
            if (!probe(hw, off)) walk(hw, off, 0u);
        }
        lds_barrier();
        for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) {
            gk4[s] = lk4[s];
            if (in_mode != 1) lk4[s] = make_uint4(~0u, ~0u, ~0u, ~0u);
        }
        lds_barrier();
    }
    if (full) atomicOr(err, DERR_TABLE_FULL);
}

// wide slots: keys + counts of the region in LDS (48 KiB: three workgroups per CU)
template <int IT>
__global__ void __launch_bounds__(BUILD_THREADS) k_part_build(const uint64_t *__restrict__ items, const uint64_t *__restrict__ leafstart,
                                                              uint32_t n_regions, CountTable t, int in_mode, uint32_t *err,
                                                              uint64_t leaf_stride, const uint32_t *__restrict__ leafcnt) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t R = t.rmask + 1; // >= 1024
    uint64_t *lk = reinterpret_cast<uint64_t *>(smem);
    uint32_t *lc = reinterpret_cast<uint32_t *>(lk + R);
    uint4 *lk4 = reinterpret_cast<uint4 *>(lk), *lc4 = reinterpret_cast<uint4 *>(lc);
    const uint32_t tid = threadIdx.x;
    const int xs = 32 - t.b1, os = 32 - t.rbits;
    uint32_t full = 0;
    for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
        const uint64_t gbase = (uint64_t) r * R;
        uint64_t i0, i1;
        leaf_range(r, leafstart, leaf_stride, leafcnt, i0, i1);
        uint64_t pre_it[BUILD_PRE];
#pragma unroll
        for (int q = 0; q < BUILD_PRE; q++) {
            const uint64_t i = i0 + (uint64_t) q * BUILD_THREADS + tid;
            pre_it[q] = i < i1 ? items[i] : CKEY_EMPTY;
        }
        uint4 *gk4 = reinterpret_cast<uint4 *>(t.keys + gbase), *gc4 = reinterpret_cast<uint4 *>(t.counts + gbase);
        if (in_mode == 1) {
            for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) lk4[s] = gk4[s];
            for (uint32_t s = tid; s < R / 4; s += BUILD_THREADS) lc4[s] = gc4[s];
        } else {
            for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) lk4[s] = make_uint4(~0u, ~0u, ~0u, ~0u);
            for (uint32_t s = tid; s < R / 4; s += BUILD_THREADS) lc4[s] = make_uint4(0u, 0u, 0u, 0u);
        }
        lds_barrier();
        auto insert = [&](uint64_t item) {
            const uint64_t v = IT == IT_HASH ? khash_inv(item) : item, h = IT == IT_HASH ? item : khash(item);
            uint32_t off = ((uint32_t) (h >> xs) * t.n2) >> os;
            bool done = false;
            for (uint32_t probes = 0; probes < R; probes++) {
                unsigned long long old = atomicCAS((unsigned long long *) &lk[off], (unsigned long long) CKEY_EMPTY, (unsigned long long) v);
                if (old == CKEY_EMPTY || old == v) {
                    atomicAdd(&lc[off], 1u);
                    done = true;
                    break;
                }
                off = (off + probes + 1u) & t.rmask;
            }
            if (!done) full = 1;
        };
#pragma unroll
        for (int q = 0; q < BUILD_PRE; q++)
            if (pre_it[q] != CKEY_EMPTY) insert(pre_it[q]);
        for (uint64_t i = i0 + (uint64_t) BUILD_PRE * BUILD_THREADS + tid; i < i1; i += BUILD_THREADS) {
            const uint64_t item = items[i];
            if (item != CKEY_EMPTY) insert(item);
        }
        lds_barrier();
        for (uint32_t s = tid; s < R / 2; s += BUILD_THREADS) gk4[s] = lk4[s];
        for (uint32_t s = tid; s < R / 4; s += BUILD_THREADS) gc4[s] = lc4[s];
        lds_barrier();
    }
    if (full) atomicOr(err, DERR_TABLE_FULL);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static size_t build_lds(const kmu_counter *c) { return c->qw ? (size_t) 8 << c->rbits : (size_t) 12 << c->rbits; }
static size_t build_pool_lds(const kmu_counter *c) { return c->qw ? (size_t) (BUILD_THREADS / 64) * BUILD_POOL * 8 : 0; } // (k_part_build_q's pools)
// the items of every region (leaves: the regions' bounds; or leaf_stride / leafcnt: fixed-size leaves with their fills) into the table
template <int IT>
static int launch_build(kmu_counter *c, const uint64_t *items, const uint64_t *leaves, uint32_t *d_err,
                        uint64_t leaf_stride = 0, const uint32_t *leafcnt = nullptr, bool leaf6 = false) {
    kmu_ctx *ctx = c->ctx;
    const uint64_t n_regions = table_regions(c);
    const int in_mode = c->empty ? 0 : 1;
    const size_t lds = build_lds(c) + build_pool_lds(c);
    const int per_cu = c->qw ? std::min(32 / (BUILD_THREADS / 64), (int) ((160 * 1024) / lds)) : 3;
    const int grid = (int) std::min<uint64_t>(n_regions, (uint64_t) ctx->num_cus * per_cu * 8);
    {
        KernelTimer tm(ctx, c->qw ? "k_part_build_q" : "k_part_build"); // (the kernels' own names)
        if (c->qw) {
            const auto kern = leaf6 ? k_part_build_q<IT_HASH, true> : k_part_build_q<IT, false>;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(BUILD_THREADS), lds, ctx->stream, items, leaves, (uint32_t) n_regions, table_of(c), in_mode, d_err,
                               leaf_stride, leafcnt);
        }
        else
            hipLaunchKernelGGL((k_part_build<IT>), dim3(grid), dim3(BUILD_THREADS), lds, ctx->stream, items, leaves, (uint32_t) n_regions,
                               table_of(c), in_mode, d_err, leaf_stride, leafcnt);
    }
    KMU_HIP(ctx, hipGetLastError());
    c->empty = false;
    return KMU_OK;
}

// the LDS-staged scatter kernels ask for more than 64 KiB of dynamic LDS: one attribute call per instantiation and device
// (function attributes are per device: remembered per context, not per process)
static int scatter_attrs(kmu_ctx *ctx) {
    if (ctx->lds_attr_set & 1u) return KMU_OK;
    const void *fns[] = {(const void *) k_part_scatter1_exact, (const void *) k_part_scatter1,
                         (const void *) k_arr_scatter_exact<IT_HASH>, (const void *) k_arr_scatter_exact<IT_KEY>, (const void *) k_arr_scatter_exact<IT_KEY_TO_HASH>,
                         (const void *) k_arr_scatter_seg<IT_HASH, false>, (const void *) k_arr_scatter_seg<IT_HASH, true>,
                         (const void *) k_arr_scatter_seg<IT_KEY_TO_HASH, false>, (const void *) k_smer_scatter1};
    for (const void *f : fns) KMU_HIP(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ctx->lds_attr_set |= 1u;
    return KMU_OK;
}

// the partition plan of a table: its own region map
bool part_plan_for(const kmu_counter *c, PartPlan *pl) {
    memset(pl, 0, sizeof *pl);
    pl->b1 = c->b1;
    pl->n2 = c->n2;
    return c->b1 <= 11 && c->n2 <= GROUP_REGIONS_MAX;
}

// ---- the single-pass partition -------------------------------------------------------------------------------------------
// The k-mers travel as khash(k-mer), so the digits are uniform: a stream's share of the items is items / bins with a standard
// deviation of sqrt(that).  Every stream gets a FIXED capacity of mean + 5 sigma + 32 items (1.4 % over the mean at level 1 of
// the bench size, 13 % per region leaf), the scatters run without their histogram passes (-7.6 and -6.0 ms), the unused tail of
// every level-1 stream is filled with "no k-mer" marks that level 2 skips, the leaves carry their fill in a count of their own.
// An item that finds its stream full goes to a spill list that is inserted into the finished table (seg_spill,
// k_count_add_spill); only a full spill list (k-mers that the hash cannot spread: a genome of one repeated k-mer) raises a flag
// that is read before the build touches the table: the call then takes the exact route from scratch.  Level 1 without a global
// histogram is also what lets kmu_sketch_count partition a chunk while the next one is uploaded.

// the overflow word block of a single-pass partition (see seg_spill) and its spill list: room for 1/32 of the items
static int seg_spill_setup(kmu_ctx *ctx, uint64_t n_items, void **ovf_out) {
    void *ovf, *sp;
    const uint64_t cap = std::min<uint64_t>(n_items / 32 + 4096, 0x7FFFFFFFull);
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

bool seg_partition_wanted(uint64_t n_items) {
    const char *e = getenv("KMU_COUNT_SEG"); // 0: always the exact two-pass levels; 2: also for small batches (tests)
    if (e && atoi(e) == 0) return false;
    if (e && atoi(e) == 2) return true;
    if (n_items >> 35) return false; // (positions inside a level-1 bin and the cursors of the shared streams are 32-bit numbers:
                                     //  a set of streams sees at most n / sets items, whatever their bins)
    // (with streams shared by a set's units the route wins wherever a table has two levels: 5.9 / 8.8 / 17.6 / 35 / 70 Mbases:
    //  0.22 / 0.27 / 0.40 / 0.66 / 1.14 ms against 0.65 / 0.70 / 0.83 / 1.14 / 1.68 for the exact levels, scripts/r03_segthr.sh)
    return n_items >= (1ull << 22);
}
static uint64_t seg_cap_for(double mean) {
    double pct = 1.0;
    if (const char *e = getenv("KMU_COUNT_SEG_PCT")) pct = std::max(0.01, atof(e) / 100.0); // tests: force overflows
    // 5 sigma of independent k-mers: three streams in ten million overflow, by a few items that the spill list takes
    const double cap = (mean + 5.0 * std::sqrt(mean) + 32.0) * pct;
    return ((uint64_t) cap + 15) & ~(uint64_t) 15; // whole 128-byte lines
}
// level 1: sets of shared streams, two per XCD (bench workload, same box: 15.8-16.1 ms; one per XCD 18.5, four 15.9-16.7, eight
// 20.3, one for the whole chip 19.8-20.1, a unit's own streams 17.6-21.5 in two states).  Level 2: units per level-1 bin that
// share the bin's leaves (one unit with its own leaves: 23.5-24.2 ms; shared by 1 / 2 / 4 / 8 / 16 / 32 / 64 / 128 units: 21.6 /
// 22.3 / 20.3-21.4 / 18.5 / 16.9-17.4 / 17.7-17.9 / 17.9 / 20.0)
static constexpr uint32_t SEG_SETS = 16, SEG_L2_UNITS = 16;
struct SegPlan {
    uint32_t units1, steps_per_unit, sets;
    uint64_t cap1, cap2; // items per level-1 stream (set, bin) / per leaf
};
static SegPlan seg_plan(const kmu_ctx *ctx, uint64_t total_bases, const PartPlan &pl) {
    SegPlan sp;
    const uint64_t nsteps = ((total_bases + 15) / 16 + 63) / 64;
    const uint32_t bins1 = plan_bins1(pl);
    // units of level 1: one workgroup per CU; under the upload of kmu_sketch_count the same units take a slice of every arrival
    sp.units1 = (uint32_t) std::min<uint64_t>(std::max<uint64_t>(nsteps, 1), (uint64_t) ctx->num_cus);
    sp.steps_per_unit = (uint32_t) ((std::max<uint64_t>(nsteps, 1) + sp.units1 - 1) / sp.units1);
    sp.units1 = (uint32_t) ((std::max<uint64_t>(nsteps, 1) + sp.steps_per_unit - 1) / sp.steps_per_unit);
    sp.sets = std::min(SEG_SETS, sp.units1);
    const uint64_t units_per_set = (sp.units1 + sp.sets - 1) / sp.sets;
    sp.cap1 = seg_cap_for((double) units_per_set * sp.steps_per_unit * 1024.0 / bins1);
    sp.cap2 = seg_cap_for((double) total_bases / bins1 / pl.n2);
    return sp;
}
// 6-byte leaf items instead of 8: a table whose slots keep <= 48 bits of an item (w >= 16); KMU_COUNT_LEAF6=0: tests
static bool leaf6_wanted(const kmu_counter *c) {
    bool on = true;
    if (const char *e = getenv("KMU_COUNT_LEAF6")) on = atoi(e) != 0;
    return on && c->qw >= 16 && c->n2 <= LEAF6_MAX_BINS;
}
// level 2 of a single-pass partition: A (level 1's streams: ap.seg_*) -> the leaves in B, ap.chunks units per bin sharing its leaves
static int seg_launch_level2(kmu_counter *c, const ArrPlan &ap, const void *A, void *B, uint64_t cap2, void *ovf, void *leafcnt, bool leaf6) {
    kmu_ctx *ctx = c->ctx;
    KMU_HIP(ctx, hipMemsetAsync(leafcnt, 0, (size_t) ap.nparts * ap.bins * 4, ctx->stream));
    const auto k2 = leaf6 ? k_arr_scatter_seg<IT_HASH, true> : k_arr_scatter_seg<IT_HASH, false>;
    KernelTimer tm(ctx, "k_arr_scatter");
    hipLaunchKernelGGL(k2, dim3(ap.nparts * ap.chunks), dim3(SCATTER_THREADS), seg_lds_bytes(ap.bins, leaf6), ctx->stream, (const uint64_t *) A,
                       (const uint64_t *) nullptr, ap, (uint64_t *) B, cap2, (uint32_t *) ovf, (uint32_t *) leafcnt, (const uint32_t *) c->lox);
    KMU_HIP(ctx, hipGetLastError());
    return KMU_OK;
}
// state of a single-pass partition between its level-1 launches (kmu_sketch_count runs them chunk by chunk under the upload)
struct SegRun {
    PartPlan pl;
    SegPlan sp;
    DevSeqs ds;
    uint64_t total_bases = 0;
    uint64_t steps_done = 0; // wave steps of the stream that have been through level 1
    bool l1_done = false;
    void *state = nullptr, *novalid = nullptr;
    void *A = nullptr, *B = nullptr, *ovf = nullptr, *leafcnt = nullptr;
    uint32_t *d_err = nullptr;
};
// own_buffer: the level-1 output must survive other users of the shared scratch "cnt.partA" (kmu_sketch_count's chunked form:
// the sketch of the next chunk writes its (key, weight) lists there while this partition is still being filled)
static int seg_begin(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, const PartPlan &pl_in, uint32_t *d_err, SegRun *run,
                     bool own_buffer = false) {
    kmu_ctx *ctx = c->ctx;
    run->pl = pl_in;
    run->sp = seg_plan(ctx, total_bases, pl_in);
    run->ds = ds;
    run->total_bases = total_bases;
    run->steps_done = 0;
    run->l1_done = false;
    run->d_err = d_err;
    const uint32_t bins1 = plan_bins1(pl_in);
    const uint64_t n_regions = plan_regions(pl_in);
    run->pl.units1 = run->sp.units1;
    run->pl.steps_per_unit = run->sp.steps_per_unit;
    // (+ two tiles: level 2 requests whole tiles a tile ahead: its last request of the last bin ends less than two tiles behind the bin)
    KMU_TRY(dev_buf(ctx, own_buffer ? "cnt.segA" : "cnt.partA", (size_t) bins1 * run->sp.sets * run->sp.cap1 * 8 + (size_t) 2 * TILE_ITEMS * 8 + 64, &run->A));
    KMU_TRY(dev_buf(ctx, "cnt.partB", (size_t) n_regions * run->sp.cap2 * 8 + 64, &run->B));
    KMU_TRY(dev_buf(ctx, "cnt.leafcnt", (size_t) n_regions * 4 + 64, &run->leafcnt));
    KMU_TRY(seg_spill_setup(ctx, total_bases, &run->ovf));
    KMU_TRY(dev_buf(ctx, "cnt.seg_state", (size_t) run->sp.sets * bins1 * 4 + 64, &run->state));
    KMU_HIP(ctx, hipMemsetAsync(run->state, 0, (size_t) run->sp.sets * bins1 * 4, ctx->stream));
    const uint64_t nsteps = std::max<uint64_t>(1, ((total_bases + 15) / 16 + 63) / 64);
    KMU_TRY(flat_novalid(ctx, ds, total_bases, c->p.kmer_size, nsteps * 64 + 64, "cnt.novalid", &run->novalid));
    KMU_TRY(scatter_attrs(ctx));
    return KMU_OK;
}
// level 1 for the wave steps that (with their 32-base halo) lie inside the first `bases_ready` bases of the stream and have not
// been through it: every unit takes its slice of them
static int seg_level1(kmu_counter *c, SegRun *run, uint64_t bases_ready) {
    kmu_ctx *ctx = c->ctx;
    if (run->l1_done) return KMU_OK;
    const bool last = bases_ready >= run->total_bases;
    const uint64_t steps_ready = last ? ((run->total_bases + 15) / 16 + 63) / 64 : (bases_ready >= 32 ? (bases_ready - 32) / 1024 : 0);
    const uint64_t n_new = steps_ready > run->steps_done ? steps_ready - run->steps_done : 0;
    // an arrival of less than a few tiles per wave waits for the next one (KMU_COUNT_SEG_ROUND_MIN: tests); the last launch is made in any case
    const char *mn = getenv("KMU_COUNT_SEG_ROUND_MIN");
    const uint64_t min_steps = (uint64_t) run->sp.units1 * (mn ? (uint64_t) atoi(mn) : 64u);
    if (!last && n_new < min_steps) return KMU_OK;
    const uint32_t bins1 = plan_bins1(run->pl);
    if (n_new) {
        PartPlan pl = run->pl;
        pl.steps_per_unit = (uint32_t) ((n_new + run->sp.units1 - 1) / run->sp.units1);
        KernelTimer tm(ctx, "k_part_scatter1");
        hipLaunchKernelGGL(k_part_scatter1, dim3(run->sp.units1), dim3(SCATTER_THREADS), seg_lds_bytes(bins1), ctx->stream, run->ds.bases,
                           run->ds.offsets, run->ds.n_seq, c->p.kmer_size, pl, (uint64_t *) run->A,
                           SegPlan1{run->sp.cap1, run->steps_done, run->steps_done + n_new, (uint32_t *) run->ovf, run->d_err, (uint32_t *) run->state,
                                    run->sp.sets, (const uint16_t *) run->novalid});
        KMU_HIP(ctx, hipGetLastError());
    }
    run->steps_done += n_new;
    if (last) {
        run->l1_done = true;
        hipLaunchKernelGGL(k_seg_tails, dim3(std::min<uint32_t>(run->sp.sets * bins1, (uint32_t) ctx->num_cus * 8u)), dim3(256), 0, ctx->stream,
                           (const uint32_t *) run->state, run->sp.sets * bins1, (uint32_t) run->sp.cap1, (uint64_t *) run->A);
        KMU_HIP(ctx, hipGetLastError());
    }
    return KMU_OK;
}
// the rest of level 1, level 2, the overflow flag, the build.  *taken = 0: the spill list overflowed, the table is untouched.
static int seg_finish(kmu_counter *c, SegRun *run, int *taken) {
    kmu_ctx *ctx = c->ctx;
    *taken = 0;
    KMU_TRY(seg_level1(c, run, run->total_bases));
    const uint32_t bins1 = plan_bins1(run->pl);
    const bool leaf6 = leaf6_wanted(c);
    ArrPlan ap{plan_digit2(run->pl), run->pl.n2, bins1, SEG_L2_UNITS, run->sp.sets, (uint32_t) run->sp.cap1, bins1, 0u};
    KMU_TRY(seg_launch_level2(c, ap, run->A, run->B, run->sp.cap2, run->ovf, run->leafcnt, leaf6));
    uint32_t h_ovf[2] = {0, 0}; // read before the table is touched: a full spill list leaves the call to the exact route
    KMU_HIP(ctx, hipMemcpyAsync(h_ovf, run->ovf, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_ovf[0]) return KMU_OK;
    KMU_TRY(launch_build<IT_HASH>(c, (const uint64_t *) run->B, nullptr, run->d_err, run->sp.cap2, (const uint32_t *) run->leafcnt, leaf6));
    KMU_TRY(seg_spill_add(c, run->ovf, h_ovf, run->d_err));
    *taken = 1;
    return KMU_OK;
}

// the radix-partitioned build over device-resident ASCII reads
int partitioned_add(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t *d_err) {
    kmu_ctx *ctx = c->ctx;
    PartPlan pl;
    if (!part_plan_for(c, &pl)) return fail(ctx, KMU_E_UNSUPPORTED, "table too large for the two-level partitioned build");
    const uint32_t bins1 = plan_bins1(pl), bins2 = pl.n2;
    const bool two = pl.b1 != 0;
    // the single-pass partition first, BEFORE the exact route takes its buffers: both use "cnt.partA" / "cnt.partB" with different
    // sizes, and a buffer that grows is freed and allocated anew (pointers taken earlier would dangle)
    if (two && !c->no_seg && seg_partition_wanted(total_bases)) {
        SegRun run;
        int taken = 0;
        KMU_TRY(seg_begin(c, ds, total_bases, pl, d_err, &run));
        KMU_TRY(seg_finish(c, &run, &taken));
        if (taken) return KMU_OK; // (else the spill list overflowed -- very skewed k-mers -- and nothing was touched: the exact route)
    }
    const uint64_t nsteps = ((total_bases + 15) / 16 + 63) / 64;
    uint32_t units1 = (uint32_t) std::min<uint64_t>(nsteps, (uint64_t) ctx->num_cus * 8);
    if (units1 < 1) units1 = 1;
    pl.steps_per_unit = (uint32_t) ((nsteps + units1 - 1) / units1);
    units1 = (uint32_t) ((nsteps + pl.steps_per_unit - 1) / pl.steps_per_unit);
    pl.units1 = units1;
    pl.chunks2 = two ? std::max<uint32_t>(1u, 16384u / bins1) : 1u;
    const uint32_t units2 = bins1 * pl.chunks2;
    const uint64_t n_regions = plan_regions(pl);
    void *A, *B = nullptr, *hist1, *offs1, *tot1, *binstart1, *hist2 = nullptr, *offs2 = nullptr, *leafstart = nullptr, *tot2 = nullptr;
    KMU_TRY(dev_buf(ctx, "cnt.partA", total_bases * 8 + 64, &A));
    KMU_TRY(dev_buf(ctx, "cnt.hist1", (size_t) units1 * bins1 * 4, &hist1));
    KMU_TRY(dev_buf(ctx, "cnt.offs1", (size_t) units1 * bins1 * 8, &offs1));
    KMU_TRY(dev_buf(ctx, "cnt.tot1", (size_t) bins1 * 8, &tot1));
    KMU_TRY(dev_buf(ctx, "cnt.binstart1", (size_t) (bins1 + 1) * 8, &binstart1));
    if (two) {
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
        hipLaunchKernelGGL(k_part_scatter1_exact, dim3(units1), dim3(SCATTER_THREADS), scatter_lds_bytes(bins1), ctx->stream,
                           ds.bases, ds.offsets, ds.n_seq, k, pl, (const uint64_t *) offs1, (const uint64_t *) binstart1, (uint64_t *) A);
    }
    const uint64_t *items = (const uint64_t *) A;
    const uint64_t *leaves = (const uint64_t *) binstart1;
    if (two) {
        ArrPlan ap{plan_digit2(pl), bins2, bins1, pl.chunks2, 0u, 0u, 0u, 0u};
        {
            KernelTimer tm(ctx, "k_arr_hist");
            hipLaunchKernelGGL(k_arr_hist<IT_HASH>, dim3(units2), dim3(256), bins2 * 4, ctx->stream, (const uint64_t *) A,
                               (const uint64_t *) binstart1, ap, (uint32_t *) hist2);
        }
        {
            KernelTimer tm(ctx, "k_arr_scan");
            const uint32_t T = std::min<uint32_t>(256u, 1u << (31 - __builtin_clz(std::max<uint32_t>(1u, ap.chunks)))), per_wg = 256u / T;
            hipLaunchKernelGGL(k_arr_scan_a, dim3(bins1 * ((bins2 + per_wg - 1) / per_wg)), dim3(256), 0, ctx->stream,
                               (const uint32_t *) hist2, ap, T, (uint64_t *) offs2, (uint64_t *) tot2);
            hipLaunchKernelGGL(k_arr_scan_b, dim3(bins1), dim3(256), 0, ctx->stream, (const uint64_t *) tot2,
                               (const uint64_t *) binstart1, ap, (uint64_t *) leafstart);
        }
        {
            KernelTimer tm(ctx, "k_arr_scatter");
            hipLaunchKernelGGL((k_arr_scatter_exact<IT_HASH>), dim3(units2), dim3(SCATTER_THREADS), scatter_lds_bytes(bins2), ctx->stream,
                               (const uint64_t *) A, (const uint64_t *) binstart1, ap, (const uint64_t *) offs2,
                               (const uint64_t *) leafstart, (uint64_t *) B);
        }
        items = (const uint64_t *) B;
        leaves = (const uint64_t *) leafstart;
    }
    KMU_HIP(ctx, hipGetLastError());
    return launch_build<IT_HASH>(c, items, leaves, d_err);
}

// Partition a device array of u64 keys by the digits of `pl` into its 2^b1 * n2 leaves (exact levels).  Returns the partitioned
// copy and the leaf bounds (both in context scratch buffers, valid until the next partition call).
// hashed_out: the output items are khash(key) instead of the keys (what the region builds of IT_HASH take).
static int partition_by_plan(kmu_ctx *ctx, const uint64_t *in, uint64_t n, const PartPlan &pl, const uint64_t **items_out,
                             const uint64_t **bounds_out, bool hashed_out) {
    KMU_TRY(scatter_attrs(ctx));
    void *b0;
    KMU_TRY(dev_buf(ctx, "arr.bounds0", 16, &b0));
    uint64_t h0[2] = {0, n};
    KMU_HIP(ctx, hipMemcpyAsync(b0, h0, 16, hipMemcpyHostToDevice, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream)); // h0 lives on the stack
    const uint64_t *bounds = (const uint64_t *) b0;
    const uint64_t *items = in;
    uint32_t nparts = 1;
    bool first_done = false;
    const int nlevels = pl.b1 ? 2 : 1;
    for (int level = 0; level < nlevels; level++) {
        const Digit d = nlevels == 2 && level == 0 ? plan_digit1(pl) : plan_digit2(pl);
        const uint32_t bins = nlevels == 2 && level == 0 ? 1u << pl.b1 : pl.n2;
        if (bins <= 1) continue;
        uint32_t chunks = level == 0 ? (uint32_t) std::min<uint64_t>(std::max<uint64_t>(1, n / 65536), 16384)
                                     : std::max<uint32_t>(1u, 16384u / nparts);
        ArrPlan ap{d, bins, nparts, chunks, 0u, 0u, 0u, 0u};
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
            const uint32_t T = std::min<uint32_t>(256u, 1u << (31 - __builtin_clz(std::max<uint32_t>(1u, chunks)))), per_wg = 256u / T;
            hipLaunchKernelGGL(k_arr_scan_a, dim3(nparts * ((bins + per_wg - 1) / per_wg)), dim3(256), 0, ctx->stream,
                               (const uint32_t *) hist, ap, T, (uint64_t *) offs, (uint64_t *) tot);
            hipLaunchKernelGGL(k_arr_scan_b, dim3(nparts), dim3(256), 0, ctx->stream, (const uint64_t *) tot, bounds, ap,
                               (uint64_t *) outb);
        }
        {
            KernelTimer tm(ctx, "k_arr_scatter");
            const size_t slds = scatter_lds_bytes(bins);
            if (!hashed_out)
                hipLaunchKernelGGL((k_arr_scatter_exact<IT_KEY>), dim3(units), dim3(SCATTER_THREADS), slds, ctx->stream, items, bounds, ap,
                                   (const uint64_t *) offs, (const uint64_t *) outb, (uint64_t *) outbuf);
            else if (!first_done)
                hipLaunchKernelGGL((k_arr_scatter_exact<IT_KEY_TO_HASH>), dim3(units), dim3(SCATTER_THREADS), slds, ctx->stream, items, bounds,
                                   ap, (const uint64_t *) offs, (const uint64_t *) outb, (uint64_t *) outbuf);
            else
                hipLaunchKernelGGL((k_arr_scatter_exact<IT_HASH>), dim3(units), dim3(SCATTER_THREADS), slds, ctx->stream, items, bounds, ap,
                                   (const uint64_t *) offs, (const uint64_t *) outb, (uint64_t *) outbuf);
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

// Partition a device array of u64 keys by the top `region_bits` bits of khash(key) into 2^region_bits leaves (<= 22 bits: two
// 11-bit passes): the sketch path's partition of pre-hashed values (kmu_sketch.hip)
int partition_u64(kmu_ctx *ctx, const uint64_t *in, uint64_t n, int region_bits, const uint64_t **items_out,
                  const uint64_t **bounds_out, bool hashed_out) {
    if (region_bits > 22) return fail(ctx, KMU_E_UNSUPPORTED, "too many partitions (2^%d)", region_bits);
    PartPlan pl;
    memset(&pl, 0, sizeof pl);
    pl.b1 = region_bits <= 11 ? 0 : (region_bits + 1) / 2;
    pl.n2 = 1u << (region_bits - pl.b1);
    return partition_by_plan(ctx, in, n, pl, items_out, bounds_out, hashed_out);
}

// the single-pass partition for an ARRAY of canonical k-mers (what the owner of a key range receives in the OCCURRENCES
// route of a distributed add): level 1 cuts the array into chunks, sets of shared streams, no histograms
// recs != nullptr: the input is n_rec super-k-mer records holding n k-mers (kmu_smer.h) instead of an array of n k-mers
int seg_partitioned_add_kmers(kmu_counter *c, const uint64_t *d_kmers, uint64_t n, const PartPlan &pl, uint32_t *d_err, int *taken,
                              const void *recs, uint64_t n_rec) {
    kmu_ctx *ctx = c->ctx;
    *taken = 0;
    const uint32_t bins1 = plan_bins1(pl);
    const uint64_t n_regions = plan_regions(pl);
    const uint32_t chunks1 = (uint32_t) ctx->num_cus; // one unit per CU: the biggest streams, the smallest margins
    const uint32_t sets = std::min(SEG_SETS, chunks1);
    const uint64_t cap1 = seg_cap_for((double) n / sets / bins1), cap2 = seg_cap_for((double) n / bins1 / pl.n2);
    void *A, *B, *ovf, *b0, *leafcnt, *cur1;
    KMU_TRY(dev_buf(ctx, "cnt.partA", (size_t) bins1 * sets * cap1 * 8 + (size_t) 2 * TILE_ITEMS * 8 + 64, &A));
    KMU_TRY(dev_buf(ctx, "cnt.partB", (size_t) n_regions * cap2 * 8 + 64, &B));
    KMU_TRY(dev_buf(ctx, "cnt.leafcnt", (size_t) n_regions * 4 + 64, &leafcnt));
    KMU_TRY(seg_spill_setup(ctx, n, &ovf));
    KMU_TRY(dev_buf(ctx, "arr.bounds0", 16, &b0));
    KMU_TRY(dev_buf(ctx, "cnt.seg_state", (size_t) sets * bins1 * 4 + 64, &cur1));
    KMU_HIP(ctx, hipMemsetAsync(cur1, 0, (size_t) sets * bins1 * 4, ctx->stream));
    hipLaunchKernelGGL(k_fill_linear, dim3(1), dim3(256), 0, ctx->stream, (uint64_t *) b0, (uint64_t) 2, n);
    KMU_TRY(scatter_attrs(ctx));
    const ArrPlan ap1{plan_digit1(pl), bins1, 1u, chunks1, 0u, 0u, 0u, sets};
    if (recs) {
        KernelTimer tm(ctx, "k_smer_scatter1");
        hipLaunchKernelGGL(k_smer_scatter1, dim3(chunks1), dim3(SCATTER_THREADS), seg_lds_bytes(bins1), ctx->stream, (const uint32_t *) recs, n_rec, c->p.kmer_size,
                           ap1, (uint64_t *) A, cap1, (uint32_t *) ovf, (uint32_t *) cur1);
    } else {
        KernelTimer tm(ctx, "k_arr_scatter");
        hipLaunchKernelGGL((k_arr_scatter_seg<IT_KEY_TO_HASH, false>), dim3(chunks1), dim3(SCATTER_THREADS), seg_lds_bytes(bins1), ctx->stream, d_kmers,
                           (const uint64_t *) b0, ap1, (uint64_t *) A, cap1, (uint32_t *) ovf, (uint32_t *) cur1, (const uint32_t *) nullptr);
    }
    hipLaunchKernelGGL(k_seg_tails, dim3(std::min<uint32_t>(sets * bins1, (uint32_t) ctx->num_cus * 8u)), dim3(256), 0, ctx->stream, (const uint32_t *) cur1,
                       sets * bins1, (uint32_t) cap1, (uint64_t *) A);
    KMU_HIP(ctx, hipGetLastError());
    const bool leaf6 = leaf6_wanted(c);
    const ArrPlan ap2{plan_digit2(pl), pl.n2, bins1, SEG_L2_UNITS, sets, (uint32_t) cap1, bins1, 0u};
    KMU_TRY(seg_launch_level2(c, ap2, A, B, cap2, ovf, leafcnt, leaf6));
    uint32_t h_ovf[2] = {0, 0};
    KMU_HIP(ctx, hipMemcpyAsync(h_ovf, ovf, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_ovf[0]) return KMU_OK; // the table is untouched: the exact levels take over
    KMU_TRY(launch_build<IT_HASH>(c, (const uint64_t *) B, nullptr, d_err, cap2, (const uint32_t *) leafcnt, leaf6));
    KMU_TRY(seg_spill_add(c, ovf, h_ovf, d_err));
    *taken = 1;
    return KMU_OK;
}

// big batches of explicit canonical k-mers (device arrays): partition by region, then the LDS build
int partitioned_add_kmers(kmu_counter *c, const uint64_t *d_kmers, uint64_t n, uint32_t *d_err) {
    kmu_ctx *ctx = c->ctx;
    PartPlan pl;
    if (!part_plan_for(c, &pl)) return fail(ctx, KMU_E_UNSUPPORTED, "table too large for the two-level partitioned build");
    if (!c->no_seg && pl.b1 && seg_partition_wanted(n)) {
        int taken = 0;
        KMU_TRY(seg_partitioned_add_kmers(c, d_kmers, n, pl, d_err, &taken));
        if (taken) return KMU_OK;
    }
    const uint64_t *items, *bounds;
    // (a table of a single region is not partitioned at all: the items stay keys)
    const bool hashed = plan_regions(pl) > 1;
    KMU_TRY(partition_by_plan(ctx, d_kmers, n, pl, &items, &bounds, hashed));
    if (hashed) return launch_build<IT_HASH>(c, items, bounds, d_err);
    return launch_build<IT_KEY>(c, items, bounds, d_err);
}

// canonical k-mers of the reads, grouped by owner rank (one level of the partition machinery with digit = owner).
// Two halves, so that a distributed add can look at the duplication sample between them: the census (per-unit histogram of
// the owners + the sample), then the scatter.
int owner_census(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t n_parts, uint32_t *d_err, OwnerPlan *op,
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
int owner_scatter(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, const OwnerPlan &op, uint64_t **dev_out) {
    kmu_ctx *ctx = c->ctx;
    void *out;
    KMU_TRY(dev_buf(ctx, "cnt.partB", total_bases * 8 + 64, &out));
    KMU_TRY(scatter_attrs(ctx));
    {
        KernelTimer tm(ctx, "k_part_scatter1");
        hipLaunchKernelGGL(k_part_scatter1_exact, dim3(op.pl.units1), dim3(SCATTER_THREADS), scatter_lds_bytes(op.pl.owner_parts), ctx->stream,
                           ds.bases, ds.offsets, ds.n_seq, c->p.kmer_size, op.pl, (const uint64_t *) op.offs1,
                           (const uint64_t *) op.binstart1, (uint64_t *) out);
    }
    KMU_HIP(ctx, hipGetLastError());
    *dev_out = (uint64_t *) out;
    return KMU_OK;
}

// distinct keys of a device list of n sampled k-mers (scratch: "cnt.sample_tab"; d_word: a device word for the count)
int sample_distinct(kmu_ctx *ctx, const uint64_t *d_list, uint32_t n, uint32_t *d_word, uint32_t *distinct_out) {
    *distinct_out = 0;
    if (!n) return KMU_OK;
    void *stab;
    uint32_t tbits = 10;
    while ((1ull << tbits) < 2ull * n) tbits++;
    KMU_TRY(dev_buf(ctx, "cnt.sample_tab", ((size_t) 8 << tbits) + 64, &stab));
    KMU_HIP(ctx, hipMemsetAsync(stab, 0xFF, (size_t) 8 << tbits, ctx->stream));
    KMU_HIP(ctx, hipMemsetAsync(d_word, 0, 4, ctx->stream));
    {
        KernelTimer tm(ctx, "k_sample_distinct");
        hipLaunchKernelGGL(k_sample_distinct, dim3(grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, d_list, n, (uint64_t *) stab,
                           (uint32_t) ((1u << tbits) - 1u), d_word);
    }
    KMU_HIP(ctx, hipGetLastError());
    KMU_HIP(ctx, hipMemcpyAsync(distinct_out, d_word, 4, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KMU_OK;
}

// occurrences / distinct of the k-mers of a batch of reads, from the key sample of a census pass (0: no estimate)
int sample_ratio(kmu_counter *c, const DevSeqs &ds, uint64_t total_bases, uint32_t *d_err, double *ratio_out) {
    kmu_ctx *ctx = c->ctx;
    *ratio_out = 0.0;
    void *slist, *sn;
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
    uint32_t d_s = 0;
    KMU_TRY(sample_distinct(ctx, (const uint64_t *) slist, n_s, (uint32_t *) sn + 2, &d_s));
    if (d_s) *ratio_out = (double) n_s / (double) d_s;
    return KMU_OK;
}

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
    if (!partitioned || !part_plan_for(c, &pl) || !pl.b1 || !seg_partition_wanted(total_bases)) return KMU_OK;
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
    if (rc == KMU_OK && !taken) { // the spill list overflowed: the exact route (not a second attempt on the same k-mers)
        c->no_seg = true;
        rc = local_add(c, h->run.ds, h->run.total_bases, h->run.d_err);
        c->no_seg = false;
    }
    delete h;
    return rc;
}
void count_chunked_abort(void *handle) { delete (CountChunked *) handle; }

} // namespace kmu
