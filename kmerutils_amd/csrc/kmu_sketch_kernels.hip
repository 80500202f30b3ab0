// kmu_sketch_kernels.hip -- per-sequence sketching kernels (ProbMinHash3a / SuperMinHash / bottom-k) for gfx950.
//
// Reference loop being replaced (src/sketching/seqsketchjaccard.rs:224-243, setsketchert.rs:121-157):
//     for every read (rayon):  FnvHashMap<Val,u64> of fhash(kmer) over all k-mers  ->  ProbMinHash3a(m)
// MI355X mapping: one persistent workgroup per CU pulls reads from an atomic queue.  The read's weighted
// multiset is built in LDS by a counting sort on a 12-bit hash bucket: every k-mer takes a rank in its bucket with
// one ds_add_rtn, an in-place scan turns the bucket counts into starts, the keys are placed densely (dk[], dw[] = 1)
// and every key then looks for an earlier equal key inside its own (short) bucket segment -- a repeat zeroes its own
// weight and adds one to the first occurrence.  No compare-and-swap probing: the divergent probe loop of a hash
// table cost ~250 wave instructions per 64 k-mers on this VALU-bound kernel.  Reads with more k-mers than the dense
// arrays hold (~10.6 k) are processed in P hash-partitions (a key always lands in one partition, so counts stay
// exact).  The m slot minima (h as order-preserving f64 bits, arg-min key) stay in LDS across passes.
// Integer / f64 ALU + LDS only; HBM traffic = the read's bases in, m signatures out.
#include <algorithm>
#include <cmath>

#define KMU_SKETCH_KERNELS_TU
#include "kmu_sketch_kernels.h"
#include "kmu_stream.h"

namespace kmu {

static constexpr uint64_t H_INIT = 0x7FEFFFFFFFFFFFFFull;    // bits of f64::MAX (MaxValueTracker initial value)
static constexpr uint64_t H_BUSY = 0xFFFFFFFFFFFFFFFEull;    // slot being updated


__device__ __forceinline__ uint32_t mix32(uint64_t key) {
    uint32_t x = (uint32_t) key ^ (uint32_t) (key >> 32);
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    return x;
}
__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t) (((uint64_t) a * b) >> 32); }



// slot update: keep (h, key) minimal per slot; exact ties go to the smaller key (order independence)
__device__ __forceinline__ void slot_update(uint64_t *hmin, uint64_t *sig, uint32_t k, double h, uint64_t key) {
    const uint64_t hb = (uint64_t) __double_as_longlong(h);
    for (;;) {
        uint64_t cur = __hip_atomic_load(&hmin[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == H_BUSY) continue;
        if (hb > cur) return;
        if (hb == cur && key >= __hip_atomic_load(&sig[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
        if (atomicCAS((unsigned long long *) &hmin[k], (unsigned long long) cur, (unsigned long long) H_BUSY) == cur) {
            __hip_atomic_store(&sig[k], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __threadfence_block();
            atomicExch((unsigned long long *) &hmin[k], (unsigned long long) hb);
            return;
        }
    }
}

// the same for slot arrays that belong to ONE wave (k_pmh_points): the lanes of a call run in lock step, so the minimum
// is taken by one LDS atomic and the winner is whoever finds its own value there afterwards; no lock word, no loop.
// Lanes of a call that meet in a slot with the same h (and therefore the same `cur`) take the same branch below.
__device__ __forceinline__ void slot_update_wave(uint64_t *hmin, uint64_t *sig, uint32_t k, double h, uint64_t key) {
    const uint64_t hb = (uint64_t) __double_as_longlong(h);
    const uint64_t cur = __hip_atomic_load(&hmin[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const bool cand = hb <= cur;
    if (cand) __hip_atomic_fetch_min(&hmin[k], hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (cand && __hip_atomic_load(&hmin[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == hb) {
        if (hb < cur) __hip_atomic_store(&sig[k], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // the old key is obsolete
        __hip_atomic_fetch_min(&sig[k], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);           // exact ties: smaller key
    }
}

// q_max = max over slots of the current minima (MaxValueTracker root); a slot in flight counts as "unknown" = MAX
__device__ __forceinline__ uint64_t wave_qmax(const uint64_t *hmin, int m) {
    uint64_t q = 0;
    for (int i = lane_id(); i < m; i += 64) {
        uint64_t v = __hip_atomic_load(&hmin[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v == H_BUSY) v = H_INIT;
        q = v > q ? v : q;
    }
    return wave_max_u64(q);
}

__device__ __forceinline__ uint32_t draw_slot(const SketchArgs &a, Xoshiro &rng) {
    if (a.rand08) {
        for (;;) {
            // v * m as 96 bits (m < 2^32): two 32 x 32 -> 64 multiply-adds instead of a full 64 x 64 high product
            const uint64_t v = rng.next();
            const uint64_t p0 = (uint64_t) (uint32_t) v * (uint32_t) a.m;
            const uint64_t p1 = (uint64_t) (uint32_t) (v >> 32) * (uint32_t) a.m + (p0 >> 32);
            const uint64_t lo = (p1 << 32) | (uint32_t) p0;
            if (lo <= a.idx_zone) return (uint32_t) (p1 >> 32);
        }
    }
    for (;;) {
        uint64_t mm = (uint64_t) rng.next_u32() * (uint32_t) a.m;
        if ((uint32_t) mm >= a.idx_thresh) return (uint32_t) (mm >> 32);
    }
}

__device__ __forceinline__ uint64_t splitmix_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + i * 0x9e3779b97f4a7c15ull; // SplitMix64 is counter based: output i depends on seed + i*G only
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// the rejection part of ExpRestricted01::sample (reached with probability 1 - 1/c1)
__device__ __forceinline__ double exp01_rest(const Exp01 &e, Xoshiro &rng) {
    for (;;) {
        double x = rng.unif01();
        if (x < e.c2) return x;
        double y = 0.5 * rng.unif01();
        if (y > 1.0 - x) {
            x = 1.0 - x;
            y = 1.0 - y;
        }
        if (x <= e.c3 * (1.0 - y)) return x;
        if (e.c1 * y <= 1.0 - x) return x;
        if (y * e.c1 * e.lambda <= expm1_small(e.lambda * (1.0 - x))) return x;
    }
}

// q_max of the wave's slots is recomputed every 16 chunks of 64 keys (every 4: 21.0 ms, 8: 20.0, 16: 19.8 on the ONT
// workload; a stale bound only lets a few more keys into the expensive half)
static constexpr uint32_t PTS_REFRESH_MASK = 15u;
// the single kernel: wave w recomputes the workgroup's q_max when (chunk + w) % 16 == 0, i.e. one of the sixteen waves per
// chunk of 1024 keys, and posts it for the others (every 4: 88.7 ms, 8: 87.3, 16: 87.0 on the ONT workload)
static constexpr uint32_t B1_REFRESH_MASK = 15u;
__device__ __forceinline__ double winv_of(const double *lut, uint32_t w) {
    if (lut && w < WINV_LUT) return lut[w];
    return 1.0 / (double) w;
}

// ProbMinHash3a, pass B1: the FIRST point of every key (h1 = winv * Exp01, slot k1).  Like the crate's first loop over
// the map, a key that may need further points (winv < q_max) is only remembered (return value) -- the crate pushes it
// to `to_be_processed` and comes back to it after every key had its first point, when q_max is small and most of
// those keys are dropped without drawing anything.  `qmax` is any upper bound of the current q_max (shared word,
// refreshed now and then); pruning with a stale bound never changes the arg-min.
// The first xoshiro256++ output needs only state words s0 and s3 (= SplitMix64 outputs 1 and 4 of the seed): the
// other two are computed only for the keys whose first point survives the q_max test.
__device__ __forceinline__ bool pmh3a_first_point(const SketchArgs &a, bool sig32, uint64_t *hmin, uint64_t *sig,
                                                  uint64_t *qmax_sh, bool refresh, bool have, uint64_t key, uint32_t w,
                                                  const double *winv_lut = nullptr) {
    uint64_t qb;
    if (refresh) {
        qb = wave_qmax(hmin, a.m);
        if (lane_id() == 0) atomicMin((unsigned long long *) qmax_sh, (unsigned long long) qb);
    } else {
        qb = __hip_atomic_load(qmax_sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    bool deferred = false;
    if (have) {
        const uint64_t seed = hasher_finish(KMU_HASHER_NOHASH, key, sig32);
        const double winv = winv_of(winv_lut, w);
        Xoshiro rng;
        rng.s0 = splitmix_at(seed, 1);
        rng.s3 = splitmix_at(seed, 4);
        const uint64_t r1 = rotl64(rng.s0 + rng.s3, 23) + rng.s0;
        const double u1 = __longlong_as_double((long long) ((r1 >> 12) | 0x3FF0000000000000ull)) - 1.0;
        double x = a.e01.c1 * u1;
        const double qmax = __longlong_as_double((long long) qb);
        const bool slow = !(x < 1.0);
        if (slow || winv * x < qmax) {
            rng.s1 = splitmix_at(seed, 2);
            rng.s2 = splitmix_at(seed, 3);
            (void) rng.next(); // the draw already used
            if (slow) x = exp01_rest(a.e01, rng);
            const double h = winv * x;
            if (h < qmax) {
                uint32_t k = draw_slot(a, rng);
                slot_update(hmin, sig, k, h, key);
                deferred = winv < qmax; // the crate: `if winv < qmax { to_be_processed.push(..) }`
            }
        }
    }
    return deferred;
}

// pass B2: further points (rounds i >= 2) of the remembered keys, against the q_max reached after all first points.
// The RNG stream of a key is replayed from its seed: round 1 consumed the Exp01 draws and one slot draw.
// `qb` (bits of a q_max upper bound) is carried by the wave across calls and refreshed after every round.
template <bool WAVE_PRIVATE = false>
__device__ __forceinline__ void pmh3a_more_points(const SketchArgs &a, bool sig32, uint64_t *hmin, uint64_t *sig,
                                                  uint64_t &qb, bool alive, uint64_t key, double winv) {
    Xoshiro rng;
    uint32_t i = 2;
    if (alive) {
        rng.seed(hasher_finish(KMU_HASHER_NOHASH, key, sig32));
        (void) exp01_sample(a.e01, rng);
        (void) draw_slot(a, rng);
    }
    while (__any(alive)) {
        if (alive) {
            double qmax = __longlong_as_double((long long) qb);
            double hbase = winv * (double) (i - 1);
            if (!(hbase < qmax)) {
                alive = false;
            } else {
                double x = exp01_sample(a.e01, rng);
                double h = hbase + winv * x;
                uint32_t k = draw_slot(a, rng); // rounds >= 2 always draw the slot
                if (h < qmax) {
                    if (WAVE_PRIVATE) slot_update_wave(hmin, sig, k, h, key);
                    else slot_update(hmin, sig, k, h, key);
                }
                if (!(winv * (double) i < qmax)) alive = false;
                i++;
            }
        }
        qb = wave_qmax(hmin, a.m);
    }
}

// The per-workgroup scratch lists live in global memory and are re-used read after read: a plain load can be served
// by a stale line of this CU's vector L1 (stores write through to L2 without refreshing it), so every read of them
// bypasses L1 (agent-scope load, `sc1`).
template <typename T>
__device__ __forceinline__ T ld_scr(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... and every write is a write-through store (`sc1`), completed (vmcnt(0)) by the workgroup barrier that precedes
// the reads: the "sc1 stores and loads on both sides" hand-off form of the CDNA guide.
template <typename T>
__device__ __forceinline__ void st_scr(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// A word every thread reads from the same LDS address is the same in all lanes, but the compiler cannot know: taking it
// through readfirstlane puts it (and every loop bound, address and branch derived from it) on the scalar unit.
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    return ((uint64_t) uniform_u32((uint32_t) (v >> 32)) << 32) | uniform_u32((uint32_t) v);
}

// in-place exclusive scan of bst[0..NBUCKETS); bst[NBUCKETS] = total.  wtot: one word per wave.
// Four waves do it, sixteen counters per thread moved as 16-byte LDS words: the scan is pure bookkeeping that every
// pass pays, and with all sixteen waves on it the instruction count is four times higher for the same LDS traffic
// (the other waves simply wait at the barrier).  bst must be 16-byte aligned.
__device__ __forceinline__ void bucket_scan(uint32_t *bst, uint32_t *wtot) {
    static_assert(NBUCKETS == 4096, "256 threads x 16 counters");
    const int tid = threadIdx.x;
    uint4 c[4];
    uint32_t sum = 0, incl = 0;
    if (tid < 256) {
        const uint4 *src = reinterpret_cast<const uint4 *>(bst) + 4 * tid;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            c[q] = src[q];
            sum += c[q].x + c[q].y + c[q].z + c[q].w;
        }
        incl = wave_incl_scan_u32(sum);
        if (lane_id() == 63) wtot[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t run = incl - sum;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t v = wtot[w];
            run += w < (tid >> 6) ? v : 0u;
        }
        uint4 *dst = reinterpret_cast<uint4 *>(bst) + 4 * tid;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint4 o;
            o.x = run; run += c[q].x;
            o.y = run; run += c[q].y;
            o.z = run; run += c[q].z;
            o.w = run; run += c[q].w;
            dst[q] = o;
        }
        if (tid == 255) bst[NBUCKETS] = run;
    }
    __syncthreads();
}

// the same economy for wiping the counters: four waves, 16-byte stores
__device__ __forceinline__ void bucket_clear(uint32_t *bst) {
    const int tid = threadIdx.x;
    if (tid < 256) {
        uint4 *dst = reinterpret_cast<uint4 *>(bst) + 4 * tid;
#pragma unroll
        for (int q = 0; q < 4; q++) dst[q] = make_uint4(0u, 0u, 0u, 0u);
        if (tid == 0) bst[NBUCKETS] = 0u;
    }
}

// One workgroup = one read at a time (all blocks of it in block mode).
// BOTTOMK = false: ProbMinHash3a on the multiset.  BOTTOMK = true: the multiset of hasher(fhash(kmer)) is sorted by
// the top bits of the hash itself, so the `m` smallest distinct hashes sit in the leading buckets; their exact rank
// (= output position) is "distinct keys in earlier buckets + smaller distinct keys in the own bucket".
//
// A partition pass normally sorts all its k-mer occurrences at once (SINGLE).  If the occurrences do not fit the dense
// arrays -- repetitive reads: poly-A, tandem repeats -- the pass is redone in ROUNDS of cap/2 positions; after every
// round the distinct (key, weight) pairs are compacted into a carry list that joins the next round's sort with its
// weights.  If even the distinct keys do not fit, the block is restarted with twice as many partitions.
// (A variant with the closure and k-mer type as template constants was tried: the hashing loop gets 18 % shorter, but
// the allocator then spills loop-carried state around the read header and the kernel as a whole is slower.)
// EMIT: stop after the multiset and write the distinct (key, weight) pairs of the read to global lists (k_pmh_points
// generates the points from there, one wave per read at full occupancy) instead of running pass B here.
// PLAIN: whole unpacked sequences to signature rows (the throughput case): the packed-input, block and partial-row paths
// are compiled out of that instantiation.  (Fixing the closure and the k-mer type as well was measured again on top of
// it: 93.7 against 89.1 ms -- the allocator trades the shorter hashing code for spills elsewhere.)
template <bool AA, bool BOTTOMK, bool EMIT, bool PLAIN>
__global__ void __launch_bounds__(1024) k_sketch_pmh3a(SketchArgs a) {
    if constexpr (PLAIN) { // the compiler sees constants wherever these are read below
        a.packed = 0;
        a.block_size = 0;
        a.part_h = nullptr;
        a.part_k = nullptr;
        a.packed_offsets = nullptr;
    }
    const KmerCfg cfg = a.cfg;
    const bool sig32 = a.sig_bytes == 4;
    // the headline's closure (canonical Kmer64bit through int64_hash) without the walk through apply_fhash's cases per key (see k_multiset_uq)
    const bool fast64 = !AA && cfg.fhash == KMU_FHASH_CANON_INVHASH && cfg.kmer_type == KMU_KMER64BIT;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t cap = a.cap;
    uint64_t *dk = reinterpret_cast<uint64_t *>(smem); // dense keys of the current pass, grouped by bucket
    uint64_t *hmin = dk + cap;
    uint64_t *sig = hmin + a.m;
    uint32_t *dw = reinterpret_cast<uint32_t *>(sig + a.m); // weights (0 = repeat of an earlier entry)
    uint32_t *bst = dw + cap;                                // NBUCKETS + 1: counts, then starts
    uint32_t *misc = bst + NBUCKETS + 1;
    misc += (8 - ((NBUCKETS + 1) & 7)) & 7; // keep the u64 at misc[M_QMAX] 8-byte aligned
    uint32_t *wtot = misc + M_WORDS;
    uint32_t *defc = wtot + 16; // keys set aside for partition p + 1 (DEF_PARTS counters)
    uint32_t *words = defc + DEF_PARTS;
    words += (4 - ((uintptr_t) words >> 2 & 3)) & 3; // 16-byte aligned: raw chunks are parked here as uint4
    uint64_t *qmax_sh = reinterpret_cast<uint64_t *>(&misc[M_QMAX]);
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int wave = tid >> 6, nwaves = nthreads >> 6;
    const int k = cfg.k;
    const uint32_t tile_pos = (a.tile_words - 2) * 16; // k-mer start positions covered by one staged tile
    uint64_t *scr_keys = a.scr_keys + (uint64_t) blockIdx.x * cap;
    uint32_t *scr_info = a.scr_info + (uint64_t) blockIdx.x * cap;
    uint32_t *scr_w = a.scr_w + (uint64_t) blockIdx.x * cap;
    uint64_t *def_keys = a.def_keys + (uint64_t) blockIdx.x * DEF_CAP;
    // bottom-k: the running list of the m smallest hashes re-uses the LDS of the (unused) slot minima
    uint64_t *bk_keys = hmin;
    uint32_t *bk_cnt = reinterpret_cast<uint32_t *>(sig);
    uint32_t bk_n = 0; // entries of the bottom-k running list (uniform)
    uint32_t emit_n = 0; // EMIT: list entries of the current read written by earlier passes (uniform)

    bucket_clear(bst);
    for (int s = tid; s < a.m; s += nthreads) { hmin[s] = H_INIT; sig[s] = 0; }
    if (tid == 0) { misc[M_NSCR] = 0; misc[M_DEF] = 0; misc[M_FLAGS] = 0; misc[M_FLAGS + 1] = 0; *qmax_sh = H_INIT; }
    // reads are taken from the global queue QCHUNK at a time (thread 0 keeps the cursor): one same-address atomic per
    // read would cap the whole grid at the L2's rate for a single address
    // The next chunk is requested while the last read of the current one is still to be handed out, so its latency is
    // never waited for.
    uint32_t q_next = 0, q_end = 0, q_pend = 0;
    bool q_pending = false;
    if (tid == 0) {
        q_next = atomicAdd(a.queue, (uint32_t) QCHUNK);
        q_end = q_next + QCHUNK;
        misc[M_READ] = q_next++;
    }
    __syncthreads();
    auto seq_of = [&](uint32_t q) -> uint32_t { return (!PLAIN && a.read_list) ? uniform_u32(a.read_list[q]) : q; };
    auto view_of = [&](uint32_t q) {
        // (the header words come back in vector registers although `r` is uniform: handing them to the scalar unit
        // keeps every length, bound and address derived from them off the vector ALU)
        const uint32_t r = seq_of(q);
        SeqView v;
        v.base = a.bases;
        v.len = uniform_u64(a.offsets[r + 1] - a.offsets[r]);
        v.packed = a.packed;
        if (a.packed) {
            v.begin = uniform_u64(a.packed_offsets[r]);
            v.total = a.total_bytes ? a.total_bytes
                                    : uniform_u64(a.packed_offsets[a.n_seq - 1] + (a.offsets[a.n_seq] - a.offsets[a.n_seq - 1] + 3) / 4);
        } else {
            v.begin = uniform_u64(a.offsets[r]);
            v.total = a.total_bytes ? a.total_bytes : uniform_u64(a.offsets[a.n_seq]);
        }
        return v;
    };
    // number of staged code words of a read's very first tile (block 0, positions from 0)
    auto first_tile_words = [&](const SeqView &v) -> uint32_t {
        const uint64_t nka = v.len >= (uint64_t) k ? v.len - k + 1 : 0;
        uint64_t pe0 = a.block_size ? (uint64_t) a.block_size : nka;
        if (pe0 > nka) pe0 = nka;
        if (pe0 == 0) return 0u;
        const uint64_t t1 = pe0 < (uint64_t) tile_pos ? pe0 : (uint64_t) tile_pos;
        const uint32_t ld = seq_lead(v);
        return (uint32_t) (((t1 - 1 + ld + (uint64_t) k - 1) >> 4) - (uint64_t) (ld >> 4) + 1) + 2;
    };
    // The NEXT read's header is fetched as soon as its index is known, and the first 16 chunks x 64 lanes x 16 waves of its
    // bases are requested while this read's duplicates are merged (A3): HBM -> LDS directly, raw, into the `words` area
    // (free from there on).  A fresh read starts without waiting for HBM.  pf_r = the read whose head sits there.
    SeqView nv;
    nv.base = a.bases; nv.begin = 0; nv.len = 0; nv.total = 0; nv.packed = a.packed;
    uint32_t flag_sel = 0; // uniform
    uint32_t nv_r = 0xFFFFFFFFu, pf_r = 0xFFFFFFFFu, pf_nw = 0;
    u32x4 raw_pf = (u32x4) (0u); // PLAIN: this thread's parked chunk of the next read
    // diagnostics (KMU_PMH_ABLATE & 256): thread 0 accumulates the clock spent in every phase of the read loop
    uint64_t ph_acc[10], ph_t = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) ph_acc[i] = 0;
    const bool ph_on = KMU_DIAG && ABL(256u) && tid == 0;
    auto phase = [&](int i) {
        if (ph_on) {
            const uint64_t t = __builtin_readcyclecounter();
            ph_acc[i] += t - ph_t;
            ph_t = t;
        }
    };
    lds_barrier();
    uint32_t r = uniform_u32(misc[M_READ]);
    if (ph_on) ph_t = __builtin_readcyclecounter();
    while (r < a.n_queue) {
        // Thread 0 takes the next read now (the atomic's latency hides under this read's work), posts it in
        // misc[M_NEXT] before the first barrier after the ranks are taken, and everybody picks it up behind that barrier.
        uint32_t r_next = 0, r_follow = 0xFFFFFFFFu;
        bool next_posted = false;
        if (tid == 0) {
            if (q_next == q_end) {
                if (!q_pending) q_pend = atomicAdd(a.queue, (uint32_t) QCHUNK);
                q_next = q_pend;
                q_end = q_pend + QCHUNK;
                q_pending = false;
            }
            r_next = q_next++;
            if (q_next == q_end && !q_pending) { // used one read from now
                q_pend = atomicAdd(a.queue, (uint32_t) QCHUNK);
                q_pending = true;
            }
        }
        const SeqView sv = nv_r == r ? nv : view_of(r);
        // positions inside a read are 32-bit from here on (half the scalar registers, half the vector instructions per
        // index computation); a single sequence of 2^31 bases or more is refused
        if (sv.len >= 0x80000000ull && tid == 0) atomicOr(a.err, DERR_TABLE_FULL);
        const uint32_t L = sv.len >= 0x80000000ull ? 0u : (uint32_t) sv.len;
        const uint32_t nk_all = L >= (uint32_t) k ? L - (uint32_t) k + 1u : 0u;
        if (L == 0 && tid == 0 && !a.hashed_bytes) atomicOr(a.err, 8u); // an empty list of pre-hashed values is fine
        if (nk_all == 0 && !a.hashed_bytes && wave_validate_seq(sv, wave, nwaves, AA))
            atomicOr(a.err, AA ? DERR_BAD_AA : DERR_NON_ACGT);
        const uint32_t lead = AA ? 0u : seq_lead(sv);
        // blocks of the read (src/sketching/seqblocksketch.rs:108-146); whole read = one block
        const uint32_t B = a.block_size ? a.block_size : (nk_all ? nk_all : 1u);
        uint32_t nblocks = a.block_size ? (uint32_t) (((uint64_t) L + B - 1) / B) : 1u;
        if (a.skip_longer && nk_all > a.skip_longer) nblocks = 0; // its row comes from the global path
        phase(0); // read header
        for (uint32_t blk = 0; blk < nblocks; blk++) {
            const uint64_t pb64 = (uint64_t) blk * B, pe64 = pb64 + B;
            const uint32_t pb = pb64 > nk_all ? nk_all : (uint32_t) pb64, pe = pe64 > nk_all ? nk_all : (uint32_t) pe64;
            const uint32_t nk = pe - pb;
            // number of hash partitions: any P with nk / P comfortably below the dense capacity will do (the multiset is
            // exact for every P), so no 64-bit division: a product with the reciprocal, rounded up
            uint32_t P = nk == 0 ? 0u : nk <= a.part_target ? 1u : (uint32_t) ((double) nk * a.inv_part_target) + 1u;
            if (ABL(64u)) P = 0;
            uint32_t bad = 0;
            bool full = false;
            bool redo = false; // uniform; PLAIN only
            // k-mer occurrences of positions [q0, q1) that belong to partition `part` take a bucket rank; the first
            // KREG * nthreads positions of a SINGLE pass keep their key in registers, the rest goes to the scratch.
            // A block that needs several partition passes is scanned (extracted, hashed) ONCE: pass 0 sets the keys of the
            // later partitions aside in a global list, the later passes read their keys from there.
            bool def_valid = false; // uniform
            for (bool block_done = (P == 0); !block_done;) {
                bool restart_block = false; // uniform
                def_valid = false;
                for (uint32_t part = 0; part < P && !restart_block; part++) {
                    bool rounds_mode = false; // uniform
                    for (bool part_done = false; !part_done;) {
                        uint64_t rk[KREG];
                        uint32_t rb[KREG];
#pragma unroll
                        for (int q = 0; q < KREG; q++) rb[q] = 0xFFFFFFFFu; // (rk[q] is read only where rb[q] names a key)

                        const uint32_t round_len = rounds_mode ? cap / 2 : nk;
                        uint32_t carry_n = 0; // distinct (key, weight) pairs carried from earlier rounds (in scr_*)
                        if (BOTTOMK && part > 0) { // the running list of the earlier partitions travels as carry
                            for (uint32_t i = tid; i < bk_n; i += nthreads) { st_scr(&scr_keys[i], bk_keys[i]); st_scr(&scr_w[i], bk_cnt[i]); }
                            carry_n = bk_n;
                            __syncthreads();
                        }
                        bool overflow = false; // uniform
                        for (uint32_t q0 = pb; q0 < pe && !overflow; q0 += round_len) {
                            const uint32_t q1 = pe - q0 > round_len ? q0 + round_len : pe;
                            const bool last_round = q1 == pe;
                            if (!PLAIN && carry_n) { // carried pairs take their ranks first (misc[M_NSCR] is 0 between passes)
                                if (tid == 0) misc[M_NSCR] = carry_n;
                                for (uint32_t i = tid; i < carry_n; i += nthreads) {
                                    const uint64_t key = ld_scr(&scr_keys[i]);
                                    const uint32_t b = BOTTOMK ? (uint32_t) (key >> a.bk_shift) & (NBUCKETS - 1)
                                                               : mix32(key) >> (32 - BUCKET_BITS);
                                    st_scr(&scr_info[i], (b << 16) | atomicAdd(&bst[b], 1u));
                                }
                                __syncthreads(); // orders the scratch stores above
                            }
                            // ---- A1: bucket ranks of the keys of this partition in [q0, q1) ------------------------
                            // this pass fills the sub-lists (one per later partition; more partitions than sub-lists: rescan)
                            const bool defer_on = !BOTTOMK && P > 1 && P <= DEF_PARTS + 1 && part == 0 && !rounds_mode;
                            const bool from_list = !BOTTOMK && part > 0 && !rounds_mode && def_valid;
                            if (defer_on) {
                                if ((uint32_t) tid < DEF_PARTS) defc[tid] = 0;
                                if (tid == 0) misc[M_DEF] = 0; // becomes 1 if a sub-list overflows
                                lds_barrier();
                            }
                            if (from_list) {
                                const uint32_t seg_n = uniform_u32(defc[part - 1]);
                                const uint64_t *seg = def_keys + (uint64_t) (part - 1) * DEF_SEG;
                                for (uint32_t i = tid; i < seg_n; i += nthreads) {
                                    const uint64_t key = ld_scr(&seg[i]);
                                    const uint32_t b = mix32(key) >> (32 - BUCKET_BITS);
                                    const uint32_t rank = atomicAdd(&bst[b], 1u);
                                    if (rank < 65536u) { // (a pass of a partitioned block parks its keys: use_park)
                                        const uint32_t si = atomicAdd(&misc[M_NSCR], 1u);
                                        if (si < (uint32_t) KREG * nthreads && si < cap) { dk[si] = key; dw[si] = (b << 16) | rank; }
                                    }
                                }
                            }
                            const uint32_t ntiles = from_list ? 0u : AA ? 1u : (uint32_t) (((uint64_t) (q1 - q0) + tile_pos - 1) >> a.tile_shift); // tile_pos is a power of two
                            for (uint32_t tile = 0; tile < ntiles; tile++) {
                                const uint32_t tp0 = AA ? q0 : q0 + tile * tile_pos;
                                const uint32_t tp1 = AA ? q1 : (q1 - tp0 > tile_pos ? tp0 + tile_pos : q1);
                                uint32_t wfirst = 0;
                                if (!AA) {
                                    wfirst = (tp0 + lead) >> 4;
                                    const uint32_t wlast = (uint32_t) (((uint64_t) tp1 - 1 + lead + (uint64_t) k - 1) >> 4);
                                    const uint32_t nw = (wlast - wfirst + 1) + 2;
                                    // the raw chunks of words [0, pf_nw) may have been parked here by the previous read
                                    const bool parked = pf_r == r && tp0 == 0 && (uint32_t) tid < pf_nw;
                                    pf_r = 0xFFFFFFFFu;
                                    u32x4 raw = (u32x4) (0u);
                                    if (PLAIN) {
                                        // the parked chunk was taken to registers behind the last barrier of the previous
                                        // read; a pass's first tile follows a barrier that every reader of `words` has
                                        // passed, so only the later tiles wait here
                                        if (parked) raw = raw_pf;
                                        if (tile != 0) lds_barrier();
                                    } else {
                                        if (parked) raw = reinterpret_cast<const u32x4 *>(words)[tid];
                                        lds_barrier(); // the previous user of `words` is done
                                    }
                                    for (uint32_t t = tid; t < nw; t += nthreads) {
                                        uint32_t b;
                                        words[t] = (parked && t == (uint32_t) tid && chunk_is_plain(sv, wfirst + t))
                                                       ? code_word_from_chunk(sv, wfirst + t, raw, b)
                                                       : load_code_word(sv, wfirst + t, b);
                                        bad |= b;
                                    }
                                    lds_barrier();
                                }
                                phase(1); // read header + code words staged
                                for (uint32_t pr = tp0; pr < tp1 && !ABL(4096u); pr += (uint32_t) KREG * nthreads) {
                                    // Where a key waits for the scan: in registers (one pass over a read that fits: its first
                                    // KREG * nthreads positions), parked unsorted in the still unused dense arrays (a pass of
                                    // a partitioned read keeps 1/P of the positions it scans), else in the global scratch.
                                    const bool use_park = !rounds_mode && P > 1 && (PLAIN || carry_n == 0);
                                    const bool use_regs = !rounds_mode && !use_park && tile == 0 && pr == tp0;
#pragma unroll
                                    for (int q = 0; q < KREG; q++) {
                                        const uint32_t p = pr + (uint32_t) q * nthreads + tid;
                                        if (p < tp1 && !ABL(32u)) {
                                            uint64_t val, rc = 0;
                                            if (AA && a.hashed_bytes) {
                                                val = a.hashed_bytes == 4
                                                          ? (uint64_t) reinterpret_cast<const uint32_t *>(a.hashed)[sv.begin + p]
                                                          : reinterpret_cast<const uint64_t *>(a.hashed)[sv.begin + p];
                                            } else if (AA) {
                                                val = 0;
                                                for (int j = 0; j < k; j++) {
                                                    uint32_t c = code_aa(sv.base[sv.begin + p + j]);
                                                    bad |= c == 0;
                                                    val = (val << 5) | c;
                                                }
                                            } else {
                                                const uint32_t qq = p + lead - 16u * wfirst;
                                                const uint32_t idx = qq >> 4, sh = (qq & 15u) * 2u;
                                                const uint64_t hi = ((uint64_t) words[idx] << 32) | words[idx + 1];
                                                const uint64_t v = (hi << sh) | (((uint64_t) words[idx + 2] << sh) >> 32);
                                                val = v >> (64 - 2 * k);
                                                rc = revcomp_val(val, k);
                                            }
                                            bool go = !ABL(4u);
                                            uint64_t key = 0;
                                            uint32_t h = 0;
                                            if (go) {
                                                key = (AA && a.hashed_bytes) ? val : fast64 ? int64_hash(rc < val ? rc : val) : apply_fhash(cfg, val, rc);
                                                if (BOTTOMK) key = hasher_finish(a.hasher, key, sig32);
                                                h = mix32(key);
                                                if (ABL(2u)) go = false;
                                                const uint32_t kp = P > 1 ? mulhi32(h * 0x85EBCA6Bu, P) : 0u;
                                                if (kp != part) {
                                                    go = false;
                                                    if (defer_on) { // its own pass will pick it up without re-hashing
                                                        const uint32_t di = atomicAdd(&defc[kp - 1], 1u);
                                                        if (di < DEF_SEG) st_scr(&def_keys[(uint64_t) (kp - 1) * DEF_SEG + di], key);
                                                        else misc[M_DEF] = 1u;
                                                    }
                                                }
                                            } else if (val == 0x1234567ull) full = true;
                                            if (go) {
                                                const uint32_t b = BOTTOMK ? (uint32_t) (key >> a.bk_shift) & (NBUCKETS - 1)
                                                                           : h >> (32 - BUCKET_BITS);
                                                const uint32_t rank = atomicAdd(&bst[b], 1u);
                                                if (rank < 65536u) { // else: the pass overflows and is redone in rounds
                                                    if (use_regs) { rk[q] = key; rb[q] = (b << 16) | rank; }
                                                    else {
                                                        const uint32_t si = atomicAdd(&misc[M_NSCR], 1u);
                                                        if (use_park) {
                                                            if (si < (uint32_t) KREG * nthreads && si < cap) { dk[si] = key; dw[si] = (b << 16) | rank; }
                                                        } else if (!PLAIN && si < cap) { st_scr(&scr_keys[si], key); st_scr(&scr_info[si], (b << 16) | rank); st_scr(&scr_w[si], 1u); }
                                                    }
                                                }
                                            } else if (h == 0x12345u) full = true;
                                        }
                                    }
                                }
                            }
                            if (tid == 0 && !next_posted) { misc[M_NEXT] = r_next; next_posted = true; }
                            __syncthreads();
                            if (defer_on) def_valid = uniform_u32(misc[M_DEF]) == 0u; // complete (every position scanned) if all fitted
                            // ---- A2: counts -> starts, dense placement ---------------------------------------------
                            phase(2); // A1
                            r_follow = uniform_u32(misc[M_NEXT]);
                            if (nv_r != r_follow && r_follow < a.n_queue) { nv = view_of(r_follow); nv_r = r_follow; }
                            // parked keys move to the registers (the barriers of the scan separate this from the placement)
                            const bool parked_pass = !rounds_mode && P > 1 && (PLAIN || carry_n == 0);
                            const uint32_t n_park = parked_pass ? uniform_u32(misc[M_NSCR]) : 0u;
                            if (parked_pass && n_park <= (uint32_t) KREG * nthreads && n_park <= cap) {
#pragma unroll
                                for (int q = 0; q < KREG; q++) {
                                    const uint32_t idx = (uint32_t) q * nthreads + tid;
                                    if (idx < n_park) { rk[q] = dk[idx]; rb[q] = dw[idx]; }
                                }

                            }
                            phase(9); // next read's header, parked keys -> registers
                            if (!ABL(128u)) bucket_scan(bst, wtot);
                            phase(3); // scan
                            const uint32_t n_keys = uniform_u32(bst[NBUCKETS]);
                            // (PLAIN: a single pass keeps every key in registers -- the host checks part_target -- and a
                            //  partitioned one parks them: the scratch lists are not used)
                            const uint32_t n_scr = (PLAIN || parked_pass) ? 0u : uniform_u32(misc[M_NSCR]);
                            if (n_keys > cap || n_scr > cap || n_park > (uint32_t) KREG * nthreads || n_park > cap) overflow = true;
                            if (!overflow && !ABL(1024u)) {
                                // (all bucket starts are requested before the first store: a load behind a store to LDS
                                // cannot be moved up by the compiler, and ten dependent round trips are the phase)
#pragma unroll
                                for (int q = 0; q < KREG; q++)
                                    if (rb[q] != 0xFFFFFFFFu) {
                                        const uint32_t b = rb[q] >> 16;
                                        rb[q] = (b << 16) | (bst[b] + (rb[q] & 0xFFFFu));
                                    }
#pragma unroll
                                for (int q = 0; q < KREG; q++)
                                    if (rb[q] != 0xFFFFFFFFu) {
                                        const uint32_t pos = rb[q] & 0xFFFFu;
                                        dk[pos] = rk[q];
                                        dw[pos] = 1u;
                                    }
                                for (uint32_t i = tid; i < n_scr; i += nthreads) { // written by this workgroup: L2 hits
                                    const uint32_t info = ld_scr(&scr_info[i]);
                                    const uint32_t b = info >> 16, pos = bst[b] + (info & 0xFFFFu);
                                    dk[pos] = ld_scr(&scr_keys[i]);
                                    dw[pos] = ld_scr(&scr_w[i]);
                                }
                            }
                            __syncthreads();
                            phase(4); // placement
                            // ---- A3: a key with an earlier equal key in its bucket segment hands its weight over ------
                            const bool do_pf = !AA && !BOTTOMK && !a.packed && !overflow && last_round && blk + 1 == nblocks &&
                                               part + 1 == P && nv_r == r_follow && r_follow < a.n_queue && !ABL(512u) &&
                                               (size_t) a.tile_words * 4 >= (size_t) nthreads * 16;
                            if (do_pf) {
                                uint32_t n = first_tile_words(nv);
                                if (n > (uint32_t) nthreads) n = (uint32_t) nthreads; // the head only
                                const uint64_t wf = seq_lead(nv) >> 4;
                                if ((uint32_t) tid < n && chunk_is_plain(nv, wf + tid))
                                    chunk16_to_lds(nv.base + (nv.begin & ~15ull) + 16 * (wf + tid),
                                                   reinterpret_cast<uint8_t *>(words) + (size_t) wave * 1024);
                                pf_r = r_follow;
                                pf_nw = n;
                            }
                            if (!overflow && !ABL(2048u)) {
                                // rb[q] becomes (own position << 16) | cursor; the walks of a thread's keys advance together,
                                // five LDS reads in flight at a time, instead of one key after the other
#pragma unroll
                                for (int q = 0; q < KREG; q++) {
                                    uint32_t v = 0u; // invalid: cursor == position == 0
                                    if (rb[q] != 0xFFFFFFFFu) v = ((rb[q] & 0xFFFFu) << 16) | bst[rb[q] >> 16];
                                    rb[q] = v;
                                }
                                static_assert(KREG % 5 == 0, "the duplicate walk advances five keys at a time");
#pragma unroll
                                for (int q0 = 0; q0 < KREG; q0 += 5) {
                                    for (;;) {
                                        bool act[5];
                                        uint64_t kq[5];
                                        bool any_act = false;
#pragma unroll
                                        for (int u = 0; u < 5; u++) {
                                            act[u] = (rb[q0 + u] & 0xFFFFu) < (rb[q0 + u] >> 16);
                                            kq[u] = act[u] ? dk[rb[q0 + u] & 0xFFFFu] : 0ull;
                                            any_act |= act[u];
                                        }
                                        if (!__any(any_act)) break;
#pragma unroll
                                        for (int u = 0; u < 5; u++)
                                            if (act[u]) {
                                                if (kq[u] == rk[q0 + u]) {
                                                    dw[rb[q0 + u] >> 16] = 0u;
                                                    atomicAdd(&dw[rb[q0 + u] & 0xFFFFu], 1u);
                                                    rb[q0 + u] = 0u; // done
                                                } else rb[q0 + u]++;
                                            }
                                    }
                                }
#pragma unroll
                                for (int q = 0; q < KREG; q++) rb[q] = 0xFFFFFFFFu; // consumed (a later round must not see them)
                                for (uint32_t i = tid; i < n_scr; i += nthreads) {
                                    const uint32_t info = ld_scr(&scr_info[i]);
                                    const uint32_t b = info >> 16, pos = bst[b] + (info & 0xFFFFu);
                                    const uint64_t key = dk[pos];
                                    for (uint32_t j = bst[b]; j < pos; j++)
                                        if (dk[j] == key) {
                                            // the weight is only written here (by its owner) and read at the end
                                            const uint32_t wpos = __hip_atomic_exchange(&dw[pos], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                            atomicAdd(&dw[j], wpos);
                                            break;
                                        }
                                }
                            }
                            __syncthreads();
                            phase(5); // A3
                            if (!PLAIN && !overflow && !last_round) {
                                // ---- compact the distinct pairs into the carry list (scr_keys / scr_w) ------------------
                                if (tid == 0) misc[M_NSCR] = 0;
                                __syncthreads();
                                for (uint32_t base = 0; base < n_keys; base += nthreads) { // uniform trip count (ballot)
                                    const uint32_t i = base + tid;
                                    const uint32_t w = i < n_keys ? dw[i] : 0u;
                                    const uint64_t cm = __ballot(w != 0u);
                                    if (cm) {
                                        const int leader = __ffsll((unsigned long long) cm) - 1;
                                        uint32_t basepos = 0;
                                        if (lane_id() == leader) basepos = atomicAdd(&misc[M_NSCR], (uint32_t) __popcll(cm));
                                        basepos = bcast_u32(basepos, leader);
                                        if (w != 0u) {
                                            const uint32_t pos = basepos + (uint32_t) __popcll(cm & ((1ull << lane_id()) - 1ull));
                                            st_scr(&scr_keys[pos], dk[i]);
                                            st_scr(&scr_w[pos], w);
                                        }
                                    }
                                }
                                __syncthreads();
                                carry_n = uniform_u32(misc[M_NSCR]);
                                if (carry_n > cap - cap / 2) overflow = true; // no room for another round of new k-mers
                                bucket_clear(bst);
                                __syncthreads();
                            }
                            if (!overflow && last_round) {
                                if (EMIT) {
                                    // ---- the pairs of this pass leave for the points kernel -------------------------------
                                    // a straight copy of the dense arrays, duplicates included with weight 0 (k_pmh_points
                                    // skips them): no compaction, no atomics.  emit_n = entries of this read so far (all
                                    // passes; at most one per k-mer, so the list of a read fits its bases' index range)
                                    const uint32_t rsq = seq_of(r); // (the general instantiation may be walking a list of reads)
                                    const uint64_t lbase = a.offsets[rsq] - a.offsets[0] + emit_n; // (a range of a larger read set)
                                    for (uint32_t i = tid; i < n_keys; i += nthreads) {
                                        a.lst_keys[lbase + i] = dk[i];
                                        a.lst_w[lbase + i] = dw[i];
                                    }
                                    emit_n += n_keys;
                                } else if (!BOTTOMK) {
                                    // ---- B1: the first point of every distinct key -------------------------------------
                                    uint32_t chunk = 0;
                                    bool any_deferred = false;
                                    for (uint32_t base = 0; base < n_keys; base += nthreads, chunk++) {
                                        const uint32_t i = base + tid;
                                        uint64_t key = 0;
                                        uint32_t w = 0;
                                        if (i < n_keys) { key = dk[i]; w = dw[i]; }
                                        const bool have = w != 0u;
                                        if (__any(have) && !ABL(1u)) {
                                            const bool deferred = pmh3a_first_point(a, sig32, hmin, sig, qmax_sh, ((chunk + wave) & B1_REFRESH_MASK) == 0u, have, key, w);
                                            if (deferred && !ABL(16u)) { dw[i] = w | 0x80000000u; any_deferred = true; }
                                        }
                                    }
                                    // ---- B2: more points for the remembered keys that still lie below q_max -----------
                                    // (a flag word in LDS, not __syncthreads_or: its library reduction brings static LDS,
                                    // which would cost the kernel its 160 KiB dynamic allocation; the word
                                    // alternates with every pass: it is cleared one pass after it was read)
                                    if (__any(any_deferred) && lane_id() == 0) misc[M_FLAGS + flag_sel] = 1u;
                                    lds_barrier();
                                    phase(6); // B1
                                    const bool run_b2 = uniform_u32(misc[M_FLAGS + flag_sel]) != 0u;
                                    flag_sel ^= 1u;
                                    if (tid == 0) misc[M_FLAGS + flag_sel] = 0u;
                                    if (run_b2 && !ABL(8u)) {
                                        uint64_t qb = wave_qmax(hmin, a.m);
                                        for (uint32_t base = 0; base < n_keys; base += nthreads) {
                                            const uint32_t i = base + tid;
                                            const uint32_t w = i < n_keys ? dw[i] : 0u;
                                            double winv = 0.0;
                                            bool alive = false;
                                            if (w & 0x80000000u) { // round 2 starts at h = winv * 1
                                                winv = 1.0 / (double) (w & 0x7FFFFFFFu);
                                                alive = winv < __longlong_as_double((long long) qb);
                                            }
                                            if (__any(alive)) pmh3a_more_points(a, sig32, hmin, sig, qb, alive, alive ? dk[i] : 0ull, winv);
                                        }
                                    }
                                } else {
                                    // ---- bottom-k selection: rank = distinct keys in earlier buckets + smaller ones in
                                    //      the own bucket ----------------------------------------------------------------
                                    uint32_t *dcnt = words; // the staged code words are no longer needed in this pass
                                    for (uint32_t b = tid; b < NBUCKETS; b += nthreads) {
                                        uint32_t d = 0;
                                        for (uint32_t j = bst[b]; j < bst[b + 1]; j++) d += dw[j] != 0u;
                                        dcnt[b] = d;
                                    }
                                    __syncthreads();
                                    bucket_scan(dcnt, wtot);
                                    const uint32_t n_distinct = uniform_u32(dcnt[NBUCKETS]);
                                    for (uint32_t i = tid; i < n_keys; i += nthreads) {
                                        if (dw[i] == 0u) continue;
                                        const uint64_t key = dk[i];
                                        const uint32_t b = (uint32_t) (key >> a.bk_shift) & (NBUCKETS - 1);
                                        uint32_t rnk = dcnt[b];
                                        if (rnk >= (uint32_t) a.m) continue;
                                        for (uint32_t j = bst[b]; j < bst[b + 1]; j++) rnk += (dw[j] != 0u) && dk[j] < key;
                                        if (rnk < (uint32_t) a.m) { bk_keys[rnk] = key; bk_cnt[rnk] = dw[i]; }
                                    }
                                    bk_n = n_distinct < (uint32_t) a.m ? n_distinct : (uint32_t) a.m;
                                    __syncthreads(); // bst / dk / dw are still being read until every thread is done
                                }
                            }
                            if (overflow || last_round) {
                                bucket_clear(bst);
                                if (tid == 0) misc[M_NSCR] = 0;
                                if (PLAIN && pf_r != 0xFFFFFFFFu) { // requested before A3: long landed
                                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                    if ((uint32_t) tid < pf_nw) raw_pf = reinterpret_cast<const u32x4 *>(words)[tid];
                                }
                                lds_barrier(); // the points are final (-> signature row); bst is clean for the next pass
                                phase(7); // B2 + clear
                            }
                        }
                        if (!overflow) part_done = true;
                        else if (PLAIN) { redo = true; restart_block = true; part_done = true; } // the general kernel's
                        else if (!rounds_mode) rounds_mode = true; // redo this partition round by round
                        else { restart_block = true; part_done = true; }
                    }
                }
                if (!restart_block) block_done = true;
                else if (PLAIN && redo) block_done = true;
                else if (P >= 65536u) { full = true; block_done = true; }
                else if constexpr (!PLAIN) {
                    // too many distinct keys per partition: start the block over with twice as many partitions
                    P *= 2;
                    for (int t = tid; t < a.m; t += nthreads) { hmin[t] = H_INIT; sig[t] = 0; }
                    if (tid == 0) *qmax_sh = H_INIT;
                    emit_n = 0; // (EMIT) the list of this read starts over
                    bk_n = 0;
                    __syncthreads();
                }
            }
            if (bad) atomicOr(a.err, AA ? DERR_BAD_AA : DERR_NON_ACGT);
            if (full) atomicOr(a.err, DERR_TABLE_FULL);
            if (BOTTOMK) {
                // rows: the m smallest distinct hashes ascending, padded with u64::MAX; counts wrap like the
                // reference's u16 / u8 (minhash.rs:87-96, :243-262)
                __syncthreads();
                for (int t = tid; t < a.m; t += nthreads) {
                    const bool have = (uint32_t) t < bk_n;
                    reinterpret_cast<uint64_t *>(a.sig_out)[(uint64_t) r * a.m + t] = have ? bk_keys[t] : 0xFFFFFFFFFFFFFFFFull;
                    if (a.counts_out) a.counts_out[(uint64_t) r * a.m + t] = have ? (bk_cnt[t] & a.bk_mask) : 0u;
                }
                bk_n = 0;
            } else if (EMIT) {
                if (tid == 0) { // the row is written by k_pmh_points (PLAIN, overflow: an empty list; the redo launch writes the row)
                    a.lst_n[seq_of(r)] = (PLAIN && redo) ? 0u : emit_n;
                    if (PLAIN && redo) a.redo_list[atomicAdd(a.queue + 56, 1u)] = r;
                }
                emit_n = 0;
            } else {
                // ---- signature of this block: arg-min key per slot, initobj (0) for an empty multiset -----------
                const uint32_t rs = seq_of(r);
                uint64_t row = a.block_rows ? a.block_rows[rs] + blk : (uint64_t) rs;
                if (PLAIN && redo) { // nothing of this sequence is kept: the general kernel sketches it from scratch
                    __syncthreads(); // (points of earlier partitions may still be in flight)
                    for (int t = tid; t < a.m; t += nthreads) { hmin[t] = H_INIT; sig[t] = 0; }
                    if (tid == 0) {
                        *qmax_sh = H_INIT;
                        a.redo_list[atomicAdd(a.queue + 56, 1u)] = r;
                    }
                } else
                for (int t = tid; t < a.m; t += nthreads) {
                    if (a.part_h) {
                        a.part_h[row * a.m + t] = hmin[t];
                        a.part_k[row * a.m + t] = sig[t];
                    } else {
                        uint64_t v = hmin[t] == H_INIT ? 0ull : sig[t];
                        if (sig32) reinterpret_cast<uint32_t *>(a.sig_out)[row * a.m + t] = (uint32_t) v;
                        else reinterpret_cast<uint64_t *>(a.sig_out)[row * a.m + t] = v;
                    }
                    hmin[t] = H_INIT;
                    sig[t] = 0;
                }
                if (tid == 0) *qmax_sh = H_INIT;
                if (PLAIN && redo) __syncthreads();
            }
            // (no barrier: the row and the slots are touched again only behind the barriers of the next pass)
        }
        if (r_follow == 0xFFFFFFFFu) { // a read without a single pass (no k-mer)
            if (tid == 0) misc[M_NEXT] = r_next;
            lds_barrier();
            r_follow = uniform_u32(misc[M_NEXT]);
            lds_barrier();
        }
        r = r_follow;
        phase(8); // row out
    }
    if (ph_on)
        for (int i = 0; i < 10; i++)
            atomicAdd(reinterpret_cast<unsigned long long *>(a.queue) + 8 + i, (unsigned long long) ph_acc[i]);
}

// the cheap half of pmh3a_first_point: can the first point of this key lie below q_max (bits `qb`)?  Needs two of the four
// SplitMix64 words and one f64 product; the rare keys whose first Exp01 draw falls in the sampler's rejection branch pass.
// UNIT_W: every key of the call has weight 1 (a chunk inside the weight-1 prefix of a list): 1 / w = 1.0 needs no look-up
template <bool UNIT_W = false>
__device__ __forceinline__ bool pmh3a_first_point_may_matter(const SketchArgs &a, bool sig32, uint64_t qb, uint64_t key,
                                                             uint32_t w, const double *winv_lut, uint64_t &s0, uint64_t &s3) {
    const uint64_t seed = hasher_finish(KMU_HASHER_NOHASH, key, sig32);
    s0 = splitmix_at(seed, 1);
    s3 = splitmix_at(seed, 4);
    const uint64_t r1 = rotl64(s0 + s3, 23) + s0;
    const double u1 = __longlong_as_double((long long) ((r1 >> 12) | 0x3FF0000000000000ull)) - 1.0;
    const double x = a.e01.c1 * u1;
    if (UNIT_W) return !(x < 1.0) || x < __longlong_as_double((long long) qb);
    return !(x < 1.0) || winv_of(winv_lut, w) * x < __longlong_as_double((long long) qb);
}

// the other half, for a key that passed pmh3a_first_point_may_matter: s0 / s3 are the two state words it computed
__device__ __forceinline__ void pmh3a_first_point_rest(const SketchArgs &a, bool sig32, uint64_t *hmin, uint64_t *sig,
                                                       const uint64_t *qmax_sh, bool have, uint64_t key, uint32_t w,
                                                       uint64_t s0, uint64_t s3, const double *winv_lut) {
    const uint64_t qb = __hip_atomic_load(qmax_sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (have) {
        const double winv = winv_of(winv_lut, w);
        const uint64_t r1 = rotl64(s0 + s3, 23) + s0;
        const double u1 = __longlong_as_double((long long) ((r1 >> 12) | 0x3FF0000000000000ull)) - 1.0;
        double x = a.e01.c1 * u1;
        const double qmax = __longlong_as_double((long long) qb);
        const bool slow = !(x < 1.0);
        if (slow || winv * x < qmax) { // (q_max may have fallen since the key was queued)
            const uint64_t seed = hasher_finish(KMU_HASHER_NOHASH, key, sig32);
            Xoshiro rng;
            rng.s0 = s0;
            rng.s3 = s3;
            rng.s1 = splitmix_at(seed, 2);
            rng.s2 = splitmix_at(seed, 3);
            (void) rng.next(); // the draw already used
            if (slow) x = exp01_rest(a.e01, rng);
            const double h = winv * x;
            if (h < qmax) slot_update_wave(hmin, sig, draw_slot(a, rng), h, key);
        }
    }
}

// ProbMinHash3a points from the (key, weight) lists of k_sketch_pmh3a<.., EMIT>: one WAVE per read, so there is no
// workgroup barrier anywhere and a CU holds as many reads in flight as its registers allow.  LDS per wave: the slot
// minima (16 m bytes) + the shared q_max word.  Pass 1 = first point of every key; pass 2 = further rounds for the keys
// with winv < q_max (a key is deferred in pass 1 exactly when winv < q_max then, and q_max only falls: re-testing
// against the settled q_max selects a subset of the deferred keys, those that can still produce a point below it).
// LONG reads (more than pts_long_t list entries; their indices are in pts_long: [0] count, [2..] indices, k_pts_long_list)
// come first and are taken by a whole WORKGROUP: its four waves walk every fourth chunk of the list with slot arrays of their
// own and the row is the per-slot minimum of the four (smaller h, then smaller key: the rule of slot_update_wave).  A wave
// prunes with the q_max of ITS minima, which is >= the q_max of the merged ones -- it only rejects points that cannot be a
// slot's minimum -- so the row is the one a single wave makes.  One wave does 4.6e4 k-mers per ms: a 200 kb read alone took
// 4.3 ms, twice what the kernel needs for a 512 MB chunk of the host leg.
__global__ void __launch_bounds__(256) k_pts_long_list(const uint32_t *lst_n, uint32_t n_seq, uint32_t thr, uint32_t *out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_seq && lst_n[r] > thr) out[2 + atomicAdd(&out[0], 1u)] = r;
}
// q_max over the minima of the four waves of a workgroup together (a long read's waves prune with it: a point at or above it
// cannot be the minimum of its slot in the merged row either; without it every wave fills all m slots from its quarter of the
// keys alone and the four make ~3x the accepted points of one wave)
__device__ __forceinline__ uint64_t wg4_qmax(const uint64_t *arrays, size_t wave_words, int m) {
    uint64_t q = 0;
    for (int t = lane_id(); t < m; t += 64) {
        uint64_t v = H_INIT;
#pragma unroll
        for (int w4 = 0; w4 < 4; w4++) {
            const uint64_t x = __hip_atomic_load(&arrays[(size_t) w4 * wave_words + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            v = x < v ? x : v;
        }
        q = v > q ? v : q;
    }
    return wave_max_u64(q);
}

// one read's points.  WG = false: by this wave alone (chunks 0, 64, 128, ...).  WG = true: by the four waves of the workgroup,
// wave w on chunks 64 w, 64 w + 256, ... with slot arrays of its own, the row = per-slot minimum of the four.
template <bool SIG32, bool WG>
__device__ __forceinline__ void pts_one_read(const SketchArgs &a, uint32_t r, uint8_t *smem, size_t wave_words, const double *winv_lut) {
    const int wave = threadIdx.x >> 6, lane = lane_id();
    constexpr bool sig32 = SIG32;
    constexpr uint32_t cstride = WG ? 256u : 64u;
    const uint32_t cstart = WG ? 64u * (uint32_t) wave : 0u;
    uint64_t *arrays = reinterpret_cast<uint64_t *>(smem);
    uint64_t *hmin = arrays + (size_t) wave * wave_words;
    uint64_t *sig = hmin + a.m;
    uint64_t *qmax_sh = sig + a.m;
    uint64_t *qk = qmax_sh + 2;
    uint32_t *qw = reinterpret_cast<uint32_t *>(qk + 128);
    uint64_t *qs0 = qk + 128 + 64, *qs3 = qs0 + 128; // the two xoshiro state words the cheap test computed
    const uint64_t base = a.offsets[r] - a.offsets[0];
    const uint32_t n = uniform_u32(a.lst_n[r]);
    const uint32_t n_u = uniform_u32(a.lst_nu[r]); // leading entries of weight 1 without a weight word
    for (int t = lane; t < a.m; t += 64) { hmin[t] = H_INIT; sig[t] = 0; }
    if (lane == 0) *qmax_sh = H_INIT;
    if (WG) __syncthreads(); // (the other waves' arrays are looked at from the first refresh on)
    // ---- pass 1 ----
    uint32_t chunk = 0, qn = 0; // qn: queued pairs (uniform)
    uint32_t wmax = 0;          // largest weight this lane saw
    uint64_t qb = H_INIT;
    // (Round 5, measured and not kept: the keys of 2 / 4 / 8 chunks in flight instead of one -- 18.40 / 18.42 / 19.60 against 18.38 ms,
    //  profiles/r05_pts_ahead.txt.  With everything but the list walk switched off the kernel takes 9.3 ms -- 38 GB of lists at
    //  4.1 TB/s -- but the whole kernel is bound by instruction issue: 1.008e10 vector wave-instructions at the half-rate peak are
    //  18 ms, the walk's loads are under them already; profiles/r05_pts_parts.txt.)
    uint64_t key_nx = 0; // the next chunk's pair is requested one iteration ahead
    uint32_t w_nx = 1;
    {
        const uint32_t i = cstart + (uint32_t) lane;
        if (i < n) { key_nx = a.lst_keys[base + i]; w_nx = i < n_u ? 1u : a.lst_w[base + i]; }
    }
    for (uint32_t c = cstart; c < n; c += cstride, chunk++) { // uniform trip count
        const uint32_t i = c + (uint32_t) lane;
        const uint64_t key = key_nx;
        const uint32_t w = w_nx;
        const bool have = i < n && w != 0u; // weight 0: a repeat of an earlier entry
        if (have) wmax = w > wmax ? w : wmax;
        if (i + cstride < n) key_nx = a.lst_keys[base + i + cstride];
        w_nx = 1u;
        if (c + cstride + 64u > n_u) { // (uniform: the next chunk reaches beyond the weight-1 prefix)
            if (i + cstride < n && i + cstride >= n_u) w_nx = a.lst_w[base + i + cstride];
        }
        // (WG: the four waves advance through the list together, so the merged q_max is refreshed four times as often per own
        //  chunk while it still falls fast -- the first 64 own chunks -- and at the single wave's cadence per own chunk after that)
        if ((chunk & (WG && chunk < 64u ? PTS_REFRESH_MASK >> 2 : PTS_REFRESH_MASK)) == 0u) {
            qb = WG ? wg4_qmax(arrays, wave_words, a.m) : wave_qmax(hmin, a.m);
            if (lane == 0) *qmax_sh = qb;
        }
        uint64_t s0 = 0, s3 = 0;
        // (diagnostic builds, KMU_PMH_ABLATE: 16384 = no key passes and the test is not computed -- the list walk alone; 8192 = the
        //  keys that pass are queued but not worked off -- walk + cheap test; wrong rows, the parts' share of the instructions)
        const bool pass = ABL(16384u) ? false
                          : c + 64u <= n_u ? have && pmh3a_first_point_may_matter<true>(a, sig32, qb, key, w, winv_lut, s0, s3) // (uniform)
                                           : have && pmh3a_first_point_may_matter(a, sig32, qb, key, w, winv_lut, s0, s3);
        const uint64_t pm = __ballot(pass);
        if (pass) {
            const uint32_t pos = qn + (uint32_t) __popcll(pm & ((1ull << lane) - 1ull));
            qk[pos] = key;
            qw[pos] = w;
            qs0[pos] = s0;
            qs3[pos] = s3;
        }
        qn += (uint32_t) __popcll(pm);
        if (qn >= 64u) { // the newest 64
            qn -= 64u;
            if (!ABL(8192u))
            pmh3a_first_point_rest(a, sig32, hmin, sig, qmax_sh, true, qk[qn + lane], qw[qn + lane], qs0[qn + lane], qs3[qn + lane],
                                   winv_lut);
        }
    }
    if (qn) {
        const bool have = (uint32_t) lane < qn;
        if (!ABL(8192u))
        pmh3a_first_point_rest(a, sig32, hmin, sig, qmax_sh, have, have ? qk[lane] : 0ull, have ? qw[lane] : 1u, have ? qs0[lane] : 0ull,
                               have ? qs3[lane] : 0ull, winv_lut);
    }
    // ---- pass 2 ----
    // (only a key with 1 / w < q_max draws again: with the largest weight of the read at hand the lists are read a
    //  second time only where that can happen at all)
    if (WG) {
        __syncthreads(); // every wave's first points are in
        qb = wg4_qmax(arrays, wave_words, a.m);
    } else qb = wave_qmax(hmin, a.m);
    wmax = (uint32_t) wave_max_u64((uint64_t) wmax);
    if (n && wmax && winv_of(winv_lut, wmax) < __longlong_as_double((long long) qb) && !ABL(512u)) { // (ABL: diagnostic builds, pass 2 left out: wrong rows, its share of the time)
        // (a key of weight 1 draws again only while q_max > 1: with every slot hit q_max < 1 -- Exp01 is restricted to
        //  [0, 1) -- and the weight-1 prefix of the list is not read a second time)
        const uint32_t c0 = 1.0 < __longlong_as_double((long long) qb) ? 0u : (n_u & ~63u);
        uint32_t c = cstart;
        if (c < c0) c += (c0 - c + cstride - 1u) / cstride * cstride; // this wave's first chunk at or behind c0
        for (; c < n; c += cstride) {
            const uint32_t i = c + (uint32_t) lane;
            double winv = 0.0;
            bool alive = false;
            if (i < n) {
                const uint32_t w = i < n_u ? 1u : a.lst_w[base + i];
                winv = winv_of(winv_lut, w);
                alive = w != 0u && winv < __longlong_as_double((long long) qb);
            }
            if (__any(alive)) pmh3a_more_points<true>(a, sig32, hmin, sig, qb, alive, alive ? a.lst_keys[base + i] : 0ull, winv);
        }
    }
    // ---- signature row: arg-min key per slot, initobj (0) for an empty multiset ----
    if (WG) {
        __syncthreads();
        for (int t = threadIdx.x; t < a.m; t += 256) {
            uint64_t bh = arrays[t], bk = arrays[a.m + t];
#pragma unroll
            for (int w4 = 1; w4 < 4; w4++) { // smaller h, then smaller key: slot_update_wave's rule
                const uint64_t h = arrays[(size_t) w4 * wave_words + t], kk = arrays[(size_t) w4 * wave_words + a.m + t];
                if (h < bh || (h == bh && kk < bk)) { bh = h; bk = kk; }
            }
            const uint64_t v = bh == H_INIT ? 0ull : bk;
            if (sig32) reinterpret_cast<uint32_t *>(a.sig_out)[(uint64_t) r * a.m + t] = (uint32_t) v;
            else reinterpret_cast<uint64_t *>(a.sig_out)[(uint64_t) r * a.m + t] = v;
        }
        __syncthreads(); // (the arrays are wiped for the next read behind it)
    } else {
        for (int t = lane; t < a.m; t += 64) {
            const uint64_t v = hmin[t] == H_INIT ? 0ull : sig[t];
            if (sig32) reinterpret_cast<uint32_t *>(a.sig_out)[(uint64_t) r * a.m + t] = (uint32_t) v;
            else reinterpret_cast<uint64_t *>(a.sig_out)[(uint64_t) r * a.m + t] = v;
        }
    }
}

template <bool SIG32>
__global__ void __launch_bounds__(256) k_pmh_points(SketchArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = lane_id();
    // per wave: slot minima, arg-min keys, q_max word, and a queue of 128 (key, weight) pairs that passed the cheap test:
    // they are worked off 64 at a time, so the expensive half of a first point always runs with all lanes busy
    const size_t wave_words = 2 * (size_t) a.m + PTS_WAVE_WORDS;
    double *winv_lut = reinterpret_cast<double *>(reinterpret_cast<uint64_t *>(smem) + (size_t) 4 * wave_words);
    for (uint32_t t = threadIdx.x; t < WINV_LUT; t += blockDim.x) winv_lut[t] = t ? 1.0 / (double) t : 0.0;
    __syncthreads();
    // the long reads first, a workgroup each (workgroup b: entries b, b + grid, ... of the list)
    // (only where a long read would be a tail: with more than four of them per workgroup of the grid they balance among themselves
    //  as single waves' reads, and the workgroup form costs more per key -- three barriers per read, q_max over four arrays)
    uint32_t n_long = a.pts_long ? a.pts_long[0] : 0u;
    if (n_long > 4u * gridDim.x) n_long = 0u;
    for (uint32_t li = blockIdx.x; li < n_long; li += gridDim.x) pts_one_read<SIG32, true>(a, a.pts_long[2 + li], smem, wave_words, winv_lut);
    uint32_t q_next = 0, q_end = 0; // lane 0: reads are taken QCHUNK at a time
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) {
            if (q_next == q_end) {
                q_next = atomicAdd(a.queue2, (uint32_t) QCHUNK);
                q_end = q_next + QCHUNK;
            }
            r = q_next++;
        }
        r = uniform_u32(r);
        if (r >= a.n_seq) break;
        if (n_long && uniform_u32(a.lst_n[r]) > a.pts_long_t) continue; // (taken by a workgroup above)
        pts_one_read<SIG32, false>(a, r, smem, wave_words, winv_lut);
    }
}

// ---- reads that fit the registers of one workgroup: the multiset without a counting sort ------------------------------------
// In a noisy long read almost every 31-mer occurs once.  k_multiset_uq does not sort what does not need sorting: every key
// sets its bit in an occupancy bitmap A of 2^16 bits (a second hash of the key; `ds_or_rtn`), a key that finds its bit set also
// sets it in B.  After one barrier a key whose B bit is clear has PROVABLY met no equal key -- weight 1, final -- and leaves
// for the (key, weight) lists straight from the registers (nine keys in ten of an ONT read at k = 31).  The others, a few
// hundred per read, are collected in LDS and merged exactly by a miniature of the general kernel's counting sort (1 024
// buckets, rank / scan / place / walk).  No partitions, blocks, rounds, parked keys: the kernel is small, a workgroup is 512
// threads with 20 keys per thread, and TWO workgroups share a CU, so one read's barriers hide under the other's work.
// Reads with more than UQ_KEYS k-mers (or more than UQ_COLL keys in collision groups) are appended to `redo_list` and taken
// by the general list-emitting kernel in a second launch.  Output: the lists k_pmh_points reads, as k_sketch_pmh3a<EMIT>.
// Two shapes: <512 threads, 2^17-bit bitmaps, 1 024 collected keys> for reads of up to 10 240 k-mers, two workgroups per CU;
// <1024, 2^18, 2 048> for up to 20 480 k-mers, one workgroup per CU, run on the list the first shape leaves behind.
// every vector-memory request of this wave has completed (the chunks of global_load_lds have landed in LDS)
__device__ __forceinline__ void vm_wait_lds_loads() {
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0); expcnt / lgkmcnt untouched
    asm volatile("" ::: "memory");
}
// (Round 4, measured and not kept: the collision groups through an open-addressing table in LDS instead of the counting sort --
// 10.13 against 10.10 ms per launch: with two workgroups per CU the barriers of one hide under the other's key phase.)

template <int UQ_THREADS, uint32_t UQ_BM_BITS, uint32_t UQ_COLL, int MINW>
__global__ void __launch_bounds__(UQ_THREADS, MINW) k_multiset_uq(SketchArgs a) {
    typedef UqShape<UQ_THREADS, UQ_BM_BITS, UQ_COLL> SH;
    constexpr uint32_t UQ_KEYS = SH::KEYS, UQ_BM_WORDS = SH::BM_WORDS, UQ_BUCKETS = SH::BUCKETS, UQ_TILE = SH::TILE;
    static_assert((UQ_BM_WORDS / 4) % (uint32_t) UQ_THREADS == 0, "whole 16-byte stores per thread wipe a bitmap");
    static_assert(UQ_COLL % UQ_THREADS == 0 && UQ_COLL / UQ_THREADS <= 4, "collected keys per thread");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *bmA = reinterpret_cast<uint32_t *>(smem);
    uint32_t *bmB = bmA + UQ_BM_WORDS;
    uint64_t *ck = reinterpret_cast<uint64_t *>(bmB + UQ_BM_WORDS); // keys of the collision groups, as collected
    uint64_t *dk = ck + UQ_COLL;                                     // ... grouped by bucket
    uint32_t *dw = reinterpret_cast<uint32_t *>(dk + UQ_COLL);
    uint32_t *bst = dw + UQ_COLL;          // UQ_BUCKETS + 1
    uint32_t *words = bst + UQ_BUCKETS + 1; // UQ_TILE
    uint32_t *wtot = words + UQ_TILE;       // one per wave
    // [0] unique entries, [1] keys in collision groups, [4] first read, [5] the read after the current one
    uint32_t *misc = wtot + UQ_THREADS / 64;
    // the next read's chunks land here straight from HBM (global_load_lds: no register is held while they are in flight):
    // chunk t of the read at byte 16 t, i.e. lane l of the wave instruction that fetches chunks 64 j .. 64 j + 63 at 1024 j + 16 l
    // (its place as an offset from the 16-byte aligned base: a pointer that went through an integer is a FLAT pointer to the compiler --
    //  64-bit address arithmetic kept in registers the kernel does not have, spilled, and FLAT instead of LDS accesses)
    constexpr uint32_t RAW_OFF = ((8u * UQ_BM_WORDS + 20u * UQ_COLL + 4u * (UQ_BUCKETS + 1u + UQ_TILE + (uint32_t) UQ_THREADS / 64u) + 64u) + 15u) & ~15u;
    uint8_t *rawp = smem + RAW_OFF;
    const KmerCfg cfg = a.cfg;
    const int k = cfg.k, tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    for (uint32_t i = tid; i < UQ_BUCKETS + 1; i += UQ_THREADS) bst[i] = 0;
    // thread 0's cursor into the read queue: misc[8] next, [9] end of the chunk in hand, [10] the chunk asked for ahead, [11] whether one
    // is (in LDS: registers of thread 0 alone would be registers of every thread)
    auto take = [&]() -> uint32_t { // thread 0: the next queue entry (the queue is asked a read before the chunk runs out)
        if (misc[8] == misc[9]) {
            if (!misc[11]) misc[10] = atomicAdd(a.queue, (uint32_t) QCHUNK);
            misc[8] = misc[10];
            misc[9] = misc[10] + QCHUNK;
            misc[11] = 0;
        }
        const uint32_t v = misc[8]++;
        if (misc[8] == misc[9] && !misc[11]) { misc[10] = atomicAdd(a.queue, (uint32_t) QCHUNK); misc[11] = 1; }
        return v;
    };
    if (tid == 0) {
        misc[8] = misc[9] = misc[10] = misc[11] = 0;
        misc[4] = take();
        misc[5] = take(); // the header of a read is fetched TWO reads ahead: its words can then be requested a whole read ahead
    }
    __syncthreads();
    const uint64_t off_first = uniform_u64(a.offsets[0]);
    const uint64_t total = a.total_bytes ? a.total_bytes : uniform_u64(a.offsets[a.n_seq]);
    // queue entry q stands for sequence read_list[q] when a list is given (the second shape's launch), else for sequence q
    auto seq_of = [&](uint32_t q) -> uint32_t { return a.read_list ? a.read_list[q] : q; };
    uint32_t r = uniform_u32(misc[4]);
    uint32_t rs = r < a.n_queue ? uniform_u32(seq_of(r)) : 0u; // the sequence
    SeqView sv;
    sv.base = a.bases; sv.packed = 0; sv.total = total; sv.begin = 0; sv.len = 0;
    if (r < a.n_queue) { sv.begin = uniform_u64(a.offsets[rs]); sv.len = uniform_u64(a.offsets[rs + 1]) - sv.begin; }
    // A thread's register slots stand for positions in the frame of the staged words (place = position + the place of the read's
    // first base in its first word, seq_lead): slots 4 i .. 4 i + 3 = the four places of quarter (tid + UQ_THREADS x i) of the
    // words.  The four k-mers of a quarter come out of ONE window of three words, their reverse complements out of the window's
    // reverse complement (StepWin: 12 instead of 26 instructions per k-mer for extraction and reverse complement, 0.75 instead
    // of 3 LDS reads; round 4 -- before, slot q was position q x UQ_THREADS + tid.  Whole words per thread -- constant shifts,
    // 7 instructions -- leave a quarter of the threads of a typical read without a k-mer: 13.3 against 11.7 ms per launch).
    auto fits = [&](const SeqView &v) -> bool { // a read this shape takes
        const uint32_t Lv = v.len >= 0x80000000ull ? 0xFFFFFFFFu : (uint32_t) v.len;
        return Lv >= (uint32_t) k && Lv - (uint32_t) k + 1u + seq_lead(v) <= UQ_KEYS;
    };
    static_assert(UQ_KREG % 4 == 0 && UQ_THREADS % 4 == 0, "quarters of words");
    auto place_of = [&](int q) -> uint32_t { return 4u * ((uint32_t) tid + (uint32_t) UQ_THREADS * (uint32_t) (q >> 2)) + (uint32_t) (q & 3); };
    // the read after the current one: header known from the start of the current read's turn
    uint32_t r_next = uniform_u32(misc[5]);
    uint32_t rs_next = r_next < a.n_queue ? uniform_u32(seq_of(r_next)) : 0u;
    SeqView nv = sv;
    bool nv_mine = false;
    if (r_next < a.n_queue) {
        nv.begin = uniform_u64(a.offsets[rs_next]);
        nv.len = uniform_u64(a.offsets[rs_next + 1]) - nv.begin;
        nv_mine = fits(nv);
    }
    __syncthreads(); // (misc[5] is rewritten at the top of the first turn)
    // diagnostic builds (KMU_PMH_ABLATE=256): thread-0 clocks per phase -> a.queue words 8..17 (u64)
    const bool ph_on = KMU_DIAG && ABL(256u) && tid == 0;
    uint64_t ph_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = ph_on ? __builtin_readcyclecounter() : 0;
    auto phase = [&](int i) {
        if (ph_on) { const uint64_t n = __builtin_readcyclecounter(); ph_acc[i] += n - ph_t; ph_t = n; }
    };
    uint32_t pf_bad = 0; // non-ACGT bytes among this thread's words of the current read, fetched a read ahead
    bool pf_valid = false;                    // uniform
    auto n_words = [&](const SeqView &v) -> uint32_t { // staged words of a read of 1 .. UQ_KEYS k-mers (its k-mers' windows + 1)
        const uint32_t Lv = (uint32_t) v.len, ld = seq_lead(v);
        return (uint32_t) ((Lv - 1 + ld) >> 4) + 2;
    };
    while (r < a.n_queue) {
        if (tid == 0) {
            misc[5] = take(); // the read after the next one
            misc[0] = 0;
            misc[1] = 0;
        }
        const uint32_t L = sv.len >= 0x80000000ull ? 0xFFFFFFFFu : (uint32_t) sv.len;
        const uint32_t nk = L >= (uint32_t) k ? L - (uint32_t) k + 1u : 0u;
        const bool mine = nk >= 1u && nk + seq_lead(sv) <= UQ_KEYS; // else: no k-mer at all (row of zeros), or the general kernel's
        uint32_t bad = 0;
        if (L == 0 && tid == 0) atomicOr(a.err, DERR_EMPTY_SEQ);
        if (nk == 0) bad |= wave_validate_seq(sv, wave, UQ_THREADS / 64, false);
        const uint32_t lead = seq_lead(sv), wfirst = lead >> 4;
        if (mine) { // stage the read's code words (prefetched ones first), wipe the bitmaps
            const uint32_t nw = n_words(sv);
            if (pf_valid) bad |= pf_bad; // (the words are in place: written behind the last turn's key phase)
            else {
                for (uint32_t t = tid; t < nw; t += UQ_THREADS) {
                    uint32_t b;
                    words[t] = load_code_word(sv, (uint64_t) wfirst + t, b);
                    bad |= b;
                }
            }
            uint4 *za = reinterpret_cast<uint4 *>(bmA), *zb = reinterpret_cast<uint4 *>(bmB);
#pragma unroll
            for (uint32_t z = 0; z < UQ_BM_WORDS / 4 / (uint32_t) UQ_THREADS; z++) { // (2^16 bits = 512 x 16 bytes)
                za[tid + z * UQ_THREADS] = make_uint4(0u, 0u, 0u, 0u);
                zb[tid + z * UQ_THREADS] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        phase(0); // queue, staging, wipe
        lds_barrier();
        phase(1);
        const uint32_t r_nn = uniform_u32(misc[5]);
        const bool has_nn = r_nn < a.n_queue;
        // the header of the read after next: requested now, looked at at the end of this turn
        const uint32_t rs_nn = has_nn ? seq_of(r_nn) : 0u;
        const uint64_t nn_o0 = has_nn ? a.offsets[rs_nn] : 0ull, nn_o1 = has_nn ? a.offsets[rs_nn + 1] : 0ull;
        // the next read's chunks: requested now, they land in LDS under the key phase and become code words behind it
        // (round 2 requested them behind the key phase and converted them on the spot: 12 % of a read's turn in that wait)
        uint32_t nwn = 0, wfn = 0;
        if (nv_mine) {
            nwn = n_words(nv);
            wfn = seq_lead(nv) >> 4;
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const uint32_t tw = (uint32_t) tid + (uint32_t) u * UQ_THREADS;
                if (tw < nwn && chunk_is_plain(nv, (uint64_t) wfn + tw))
                    chunk16_to_lds(nv.base + (nv.begin & ~15ull) + 16 * ((uint64_t) wfn + tw), rawp + (size_t) (tw >> 6) * 1024);
            }
        }
        uint64_t rk[UQ_KREG];
        // the bitmap index of a key is a function of the key: computed again where the B bit is looked at instead of kept in
        // twenty registers (the kernel sits at its 128: 33 spilled vector registers with the indices kept)
        // (one multiplication of the folded key, by another constant than mix32's: the keys that share a bit of the bitmap must not
        //  share a bucket of the collision groups' sort; round 3's form ran the key through mix32 first: 8 instructions, twice per key)
        auto bm_index = [&](uint64_t key) -> uint32_t {
            return (((uint32_t) key ^ (uint32_t) (key >> 32)) * 0x85EBCA6Bu) >> (32 - UQ_BM_BITS);
        };
        bool over = false; // uniform: too many keys in collision groups
        if (mine) {
            // ---- keys: extract, closure, bitmaps; four positions' LDS round trips in flight at a time ----
            // (FAST: the closure of the headline -- canonical Kmer64bit through int64_hash, datasketcher.rs:225 -- without the
            //  per-key walk through apply_fhash's cases: the mode is the same for every key of the launch, and a chain of scalar
            //  compares and taken branches per key costs a workgroup of four waves per SIMD more than the arithmetic it selects)
            const uint32_t l0 = lead - 16u * wfirst; // place of the read's first base
            auto key_phase = [&](auto fast_tag) __attribute__((always_inline)) {
                constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
                for (int q0 = 0; q0 < UQ_KREG; q0 += 4) {
                    uint32_t bit[4], rbi[UQ_KREG];
                    const uint32_t wi = ((uint32_t) tid >> 2) + (uint32_t) (UQ_THREADS / 4) * (uint32_t) (q0 >> 2); // the quarter's word
                    const StepWin sw = step_win(words[wi], words[wi + 1], words[wi + 2], k);
                    // (round 4, measured and not kept: a branch-free form for the waves whose quarters are whole -- 10.6 against 10.1 ms
                    //  per launch, 19 instead of 15 spilled registers)
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int q = q0 + u;
                        rbi[q] = 0xFFFFFFFFu;
                        rk[q] = 0;
                        if (place_of(q) - l0 < nk) {
                            uint64_t val, rc;
                            step_val_rc(sw, 4u * ((uint32_t) tid & 3u) + (uint32_t) u, val, rc);
                            const uint64_t key = FAST ? int64_hash(rc < val ? rc : val) : apply_fhash(cfg, val, rc);
                            rk[q] = key;
                            rbi[q] = bm_index(key);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int q = q0 + u;
                        bit[u] = 0;
                        if (rbi[q] != 0xFFFFFFFFu) {
                            const uint32_t b = 1u << (rbi[q] & 31u);
                            bit[u] = atomicOr(&bmA[rbi[q] >> 5], b) & b;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (bit[u]) atomicOr(&bmB[rbi[q0 + u] >> 5], bit[u]);
                }
            };
            if (cfg.fhash == KMU_FHASH_CANON_INVHASH && cfg.kmer_type == KMU_KMER64BIT) key_phase(std::true_type{});
            else key_phase(std::false_type{});
            phase(2); // keys, closure, bitmaps
            lds_barrier();
            phase(3);
        }
        // ---- the next read's chunks have landed: its code words replace this read's (every key of this read is in a register by
        // now -- the barrier behind the key phase -- and nothing below looks at `words`).  Round 5: they used to wait in three registers
        // per thread until the top of the next turn, in a kernel that sits on its register limit ----
        if (nv_mine) {
            vm_wait_lds_loads();
            pf_bad = 0;
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const uint32_t tw = (uint32_t) tid + (uint32_t) u * UQ_THREADS;
                uint32_t b = 0;
                if (tw < nwn)
                    words[tw] = chunk_is_plain(nv, (uint64_t) wfn + tw)
                                    ? code_word_from_chunk(nv, (uint64_t) wfn + tw, *reinterpret_cast<const u32x4 *>(rawp + (size_t) tw * 16), b)
                                    : load_code_word(nv, (uint64_t) wfn + tw, b);
                pf_bad |= b;
            }
        }
        const uint64_t lb = sv.begin - off_first; // list entries of read r start here
        phase(4); // the next read's header and words requested
        if (mine) {
            // ---- sort out: B bit clear = occurs once = list entry (key, 1) from the register; else collect ----
            // (r03: all twenty B bits read at once and ONE atomic pair per wave instead of five -- 15.9 against 13.0 ms per launch:
            //  the kernel sits at its 128 registers, twenty more live values spill)
#pragma unroll
            for (int q0 = 0; q0 < UQ_KREG; q0 += 4) {
                bool uq[4], co[4];
                uint64_t um[4], cm[4];
                uint32_t ut = 0, ct = 0, rbi[UQ_KREG];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + u;
                    const bool have = place_of(q) - (lead - 16u * wfirst) < nk;
                    rbi[q] = bm_index(rk[q]);
                    co[u] = have && (bmB[rbi[q] >> 5] & (1u << (rbi[q] & 31u))) != 0u;
                    uq[u] = have && !co[u];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    um[u] = __ballot(uq[u]);
                    cm[u] = __ballot(co[u]);
                    ut += (uint32_t) __popcll(um[u]);
                    ct += (uint32_t) __popcll(cm[u]);
                }
                uint32_t ub = 0, cb = 0; // one atomic per wave, list and group of four register slots
                if (lane == 0) {
                    if (ut) ub = atomicAdd(&misc[0], ut);
                    if (ct) cb = atomicAdd(&misc[1], ct);
                }
                ub = bcast_u32(ub, 0);
                cb = bcast_u32(cb, 0);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint64_t below = (1ull << lane) - 1ull;
                    if (uq[u]) {
                        const uint64_t at = lb + ub + (uint32_t) __popcll(um[u] & below);
                        a.lst_keys[at] = rk[q0 + u]; // (weight 1, implied: lst_nu)
                    }
                    if (co[u]) {
                        const uint32_t at = cb + (uint32_t) __popcll(cm[u] & below);
                        if (at < UQ_COLL) ck[at] = rk[q0 + u];
                    }
                    ub += (uint32_t) __popcll(um[u]);
                    cb += (uint32_t) __popcll(cm[u]);
                }
            }
            phase(5); // sort out: unique keys to the lists, the others collected
            lds_barrier();
            phase(6);
            const uint32_t n_u = uniform_u32(misc[0]), n_c = uniform_u32(misc[1]);
            over = n_c > UQ_COLL;
            if (!over && n_c) {
                // ---- the collision groups: counting sort on 10 hash bits, equal keys hand their weight to the first ----
                uint64_t key[UQ_COLL / UQ_THREADS];
                uint32_t rb[UQ_COLL / UQ_THREADS];
#pragma unroll
                for (int j = 0; j < (int) (UQ_COLL / UQ_THREADS); j++) {
                    const uint32_t i = (uint32_t) j * UQ_THREADS + tid;
                    rb[j] = 0xFFFFFFFFu;
                    key[j] = 0;
                    if (i < n_c) {
                        key[j] = ck[i];
                        const uint32_t b = mix32(key[j]) / (0x80000000u / (UQ_BUCKETS / 2)); // the top log2(UQ_BUCKETS) bits
                        rb[j] = (b << 16) | atomicAdd(&bst[b], 1u);
                    }
                }
                lds_barrier();
                { // exclusive scan of the 1 024 bucket counts, two per thread
                    const uint32_t c0 = bst[2 * tid], c1 = bst[2 * tid + 1];
                    const uint32_t incl = wave_incl_scan_u32(c0 + c1);
                    if (lane == 63) wtot[wave] = incl;
                    lds_barrier();
                    uint32_t pre = incl - (c0 + c1);
#pragma unroll
                    for (int w = 0; w < UQ_THREADS / 64; w++) pre += w < wave ? wtot[w] : 0u;
                    bst[2 * tid] = pre;
                    bst[2 * tid + 1] = pre + c0;
                }
                lds_barrier();
#pragma unroll
                for (int j = 0; j < (int) (UQ_COLL / UQ_THREADS); j++)
                    if (rb[j] != 0xFFFFFFFFu) {
                        const uint32_t pos = bst[rb[j] >> 16] + (rb[j] & 0xFFFFu);
                        rb[j] = (rb[j] & 0xFFFF0000u) | pos;
                        dk[pos] = key[j];
                        dw[pos] = 1u;
                    }
                lds_barrier();
#pragma unroll
                for (int j = 0; j < (int) (UQ_COLL / UQ_THREADS); j++)
                    if (rb[j] != 0xFFFFFFFFu) {
                        const uint32_t pos = rb[j] & 0xFFFFu;
                        for (uint32_t t = bst[rb[j] >> 16]; t < pos; t++)
                            if (dk[t] == key[j]) { // the first equal key of the bucket takes this one's weight
                                dw[pos] = 0u;
                                atomicAdd(&dw[t], 1u);
                                break;
                            }
                    }
                lds_barrier();
                for (uint32_t i = tid; i < n_c; i += UQ_THREADS) {
                    a.lst_keys[lb + n_u + i] = dk[i];
                    a.lst_w[lb + n_u + i] = dw[i];
                }
                bst[2 * tid] = 0;
                bst[2 * tid + 1] = 0;
            }
            if (tid == 0 && !over) { a.lst_n[rs] = n_u + n_c; a.lst_nu[rs] = n_u; }
        }
        phase(7); // collision groups
        if (tid == 0) {
            if (nk == 0) a.lst_n[rs] = 0u; // no k-mer: k_pmh_points writes the row of an empty multiset
            else if (!mine || over) {       // the next kernel's: longer than the registers, or too repetitive
                a.lst_n[rs] = 0u;
                a.redo_list[atomicAdd(a.queue + 56, 1u)] = rs;
            }
        }
        if (bad) atomicOr(a.err, DERR_NON_ACGT);
        pf_valid = nv_mine;
        r = r_next;
        rs = rs_next;
        sv = nv;
        r_next = r_nn;
        rs_next = uniform_u32(rs_nn);
        nv_mine = false;
        if (has_nn) {
            nv.begin = uniform_u64(nn_o0);
            nv.len = uniform_u64(nn_o1) - nv.begin;
            nv_mine = fits(nv);
        }
        lds_barrier();
        phase(8); // end of the read's turn
    }
    if (ph_on)
        for (int i = 0; i < 10; i++)
            atomicAdd(reinterpret_cast<unsigned long long *>(a.queue) + 8 + i, (unsigned long long) ph_acc[i]);
}

// ---- reads of at most 256 k-mers (short-read sequencers): the multiset by ONE WAVE per read ---------------------------------
// A 150 bp read has ~130 k-mers: a 512-thread workgroup of k_multiset_uq spends ten barriers on a quarter of a key per thread
// (12.9 ms for a million such reads).  Here a wave takes a read by itself: the lanes stage the read's <= 20 code words in the
// wave's corner of LDS, every lane makes up to four keys, and equal keys meet in a 512-slot open-addressing table of the wave
// (`ds_cmpst_rtn_b64` claims a slot, `ds_add` counts) -- exact, no barrier, 20 waves per CU.  The occupied slots leave as the
// (key, weight) list k_pmh_points reads (its result does not depend on the order of a list).  The all-ones value that marks
// a free slot can be a key: such keys are counted in a register and listed at the end.
// Taken by launch_pmh3a when the longest read of the batch has at most SHORT_KEYS k-mers.
__global__ void __launch_bounds__(256) k_multiset_short(SketchArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    uint64_t *tk = reinterpret_cast<uint64_t *>(smem + (size_t) wave * SHORT_WAVE_BYTES);
    uint32_t *tc = reinterpret_cast<uint32_t *>(tk + SHORT_SLOTS);
    uint32_t *words = tc + SHORT_SLOTS;
    const KmerCfg cfg = a.cfg;
    const int k = cfg.k;
    const bool fast64 = cfg.fhash == KMU_FHASH_CANON_INVHASH && cfg.kmer_type == KMU_KMER64BIT; // (see k_multiset_uq)
    for (uint32_t t = (uint32_t) lane; t < SHORT_SLOTS; t += 64u) { tk[t] = ~0ull; tc[t] = 0u; }
    const uint64_t off_first = uniform_u64(a.offsets[0]);
    const uint64_t total = a.total_bytes ? a.total_bytes : uniform_u64(a.offsets[a.n_seq]);
    uint32_t q_next = 0, q_end = 0, bad = 0; // lane 0: reads are taken QCHUNK at a time
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) {
            if (q_next == q_end) {
                q_next = atomicAdd(a.queue, (uint32_t) QCHUNK);
                q_end = q_next + QCHUNK;
            }
            r = q_next++;
        }
        r = uniform_u32(r);
        if (r >= a.n_seq) break;
        SeqView sv;
        sv.base = a.bases; sv.packed = 0; sv.total = total;
        sv.begin = uniform_u64(a.offsets[r]);
        sv.len = uniform_u64(a.offsets[r + 1]) - sv.begin;
        const uint32_t L = sv.len >= 0x80000000ull ? 0xFFFFFFFFu : (uint32_t) sv.len;
        const uint32_t nk = L >= (uint32_t) k ? L - (uint32_t) k + 1u : 0u;
        if (L == 0 && lane == 0) atomicOr(a.err, DERR_EMPTY_SEQ);
        if (nk == 0) { // no k-mer: k_pmh_points writes the row of an empty multiset
            bad |= wave_validate_seq(sv, 0, 1, false);
            if (lane == 0) a.lst_n[r] = 0u;
            continue;
        }
        if (nk > SHORT_KEYS) { // (the host only sends batches whose longest read fits)
            if (lane == 0) { a.lst_n[r] = 0u; atomicOr(a.err, DERR_TABLE_FULL); }
            continue;
        }
        const uint32_t lead = seq_lead(sv), wfirst = lead >> 4;
        const uint32_t nw = (uint32_t) ((L - 1 + lead) >> 4) + 2; // the k-mers' windows + 1 (<= 20 words)
        if ((uint32_t) lane < nw) {
            uint32_t b;
            words[lane] = load_code_word(sv, (uint64_t) wfirst + (uint32_t) lane, b);
            bad |= b;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the wave's own LDS traffic is in order; this keeps the compiler from moving it
        __builtin_amdgcn_wave_barrier();
        uint32_t n_free_key = 0; // occurrences of the key that looks like a free slot (wave-uniform)
#pragma unroll
        for (int j = 0; j < (int) (SHORT_KEYS / 64); j++) {
            const uint32_t p = (uint32_t) lane + 64u * (uint32_t) j;
            const bool have = p < nk;
            uint64_t key = 0;
            if (have) {
                const uint32_t qq = p + lead - 16u * wfirst, idx = qq >> 4, sh = (qq & 15u) * 2u;
                const uint64_t hi = ((uint64_t) words[idx] << 32) | words[idx + 1];
                const uint64_t v = (hi << sh) | (((uint64_t) words[idx + 2] << sh) >> 32);
                const uint64_t val = v >> (64 - 2 * k);
                const uint64_t rc = revcomp_val(val, k);
                key = fast64 ? int64_hash(rc < val ? rc : val) : apply_fhash(cfg, val, rc);
            }
            const bool odd = have && key == ~0ull;
            n_free_key += (uint32_t) __popcll(__ballot(odd));
            if (have && !odd) {
                uint32_t slot = mix32(key) & (SHORT_SLOTS - 1);
                for (;;) { // (256 keys at most in 512 slots: a free slot always turns up)
                    const uint64_t old = atomicCAS((unsigned long long *) &tk[slot], ~0ull, (unsigned long long) key);
                    if (old == ~0ull || old == key) { atomicAdd(&tc[slot], 1u); break; }
                    slot = (slot + 1) & (SHORT_SLOTS - 1);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- the occupied slots leave as the read's list (and the table is free again) ----
        const uint64_t lb = sv.begin - off_first;
        uint32_t n_out = 0;
#pragma unroll
        for (int s8 = 0; s8 < (int) (SHORT_SLOTS / 64); s8++) {
            const uint32_t slot = (uint32_t) s8 * 64u + (uint32_t) lane;
            const uint64_t kq = tk[slot];
            const bool occ = kq != ~0ull;
            const uint64_t om = __ballot(occ);
            if (occ) {
                const uint64_t at = lb + n_out + (uint32_t) __popcll(om & ((1ull << lane) - 1ull));
                a.lst_keys[at] = kq;
                a.lst_w[at] = tc[slot];
                tk[slot] = ~0ull;
                tc[slot] = 0u;
            }
            n_out += (uint32_t) __popcll(om);
        }
        if (n_free_key) {
            if (lane == 0) { a.lst_keys[lb + n_out] = ~0ull; a.lst_w[lb + n_out] = n_free_key; }
            n_out++;
        }
        if (lane == 0) a.lst_n[r] = n_out;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (bad) atomicOr(a.err, DERR_NON_ACGT);
}

// ---- the points of a SHORT list (at most 256 pairs: the reads k_multiset_short takes) ------------------------------------------
// A read with fewer keys than m ln m cannot prune: ~m H_m points (1 200 at m = 200) are drawn before every slot is hit, round
// after round over all keys.  k_pmh_points walks a list chunk by chunk, each chunk through all of ITS rounds with the
// generator replayed from the seed, which for three chunks of a 130-key read is three times eighteen chunk-rounds; here the
// wave keeps its <= 4 pairs per lane AND their generator states in registers and takes all keys through round i before
// round i + 1 (nine rounds for the same read), q_max refreshed once per round.  Same draws per key in the same order, same
// slot arithmetic: the rows are those of k_pmh_points (the result of ProbMinHash3a does not depend on the order of the keys).
template <bool SIG32>
__global__ void __launch_bounds__(256) k_pmh_points_short(SketchArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    constexpr bool sig32 = SIG32;
    uint64_t *hmin = reinterpret_cast<uint64_t *>(smem) + (size_t) wave * (2 * (size_t) a.m + 2);
    uint64_t *sig = hmin + a.m;
    double *winv_lut = reinterpret_cast<double *>(reinterpret_cast<uint64_t *>(smem) + (size_t) 4 * (2 * (size_t) a.m + 2));
    for (uint32_t t = threadIdx.x; t < WINV_LUT; t += blockDim.x) winv_lut[t] = t ? 1.0 / (double) t : 0.0;
    __syncthreads();
    constexpr int NJ = (int) (SHORT_KEYS / 64);
    uint32_t q_next = 0, q_end = 0; // lane 0: reads are taken QCHUNK at a time
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) {
            if (q_next == q_end) {
                q_next = atomicAdd(a.queue2, (uint32_t) QCHUNK);
                q_end = q_next + QCHUNK;
            }
            r = q_next++;
        }
        r = uniform_u32(r);
        if (r >= a.n_seq) break;
        const uint64_t base = a.offsets[r] - a.offsets[0];
        const uint32_t n = a.lst_n[r]; // <= SHORT_KEYS + 1 (the all-ones key, if any, sits behind the table's pairs)
        uint64_t key[NJ + 1];
        double winv[NJ + 1];
        Xoshiro rng[NJ + 1];
        bool alive[NJ + 1];
#pragma unroll
        for (int j = 0; j <= NJ; j++) {
            const uint32_t i = (uint32_t) lane + 64u * (uint32_t) j;
            alive[j] = false;
            key[j] = 0;
            winv[j] = 0.0;
            if (i < n && (j < NJ || lane == 0)) {
                key[j] = a.lst_keys[base + i];
                const uint32_t w = a.lst_w[base + i];
                winv[j] = winv_of(winv_lut, w);
                alive[j] = w != 0u;
            }
        }
        for (int t = lane; t < a.m; t += 64) { hmin[t] = H_INIT; sig[t] = 0; }
        uint64_t qb = H_INIT;
        // ---- round 1: the first point of every key (pmh3a_first_point, with the generator kept) ----
#pragma unroll
        for (int j = 0; j <= NJ; j++) {
            if (__any(alive[j])) {
                if (alive[j]) {
                    rng[j].seed(hasher_finish(KMU_HASHER_NOHASH, key[j], sig32));
                    const double x = exp01_sample(a.e01, rng[j]);
                    const double h = winv[j] * x, qmax = __longlong_as_double((long long) qb);
                    if (h < qmax) {
                        const uint32_t k = draw_slot(a, rng[j]);
                        slot_update_wave(hmin, sig, k, h, key[j]);
                        alive[j] = winv[j] < qmax; // the crate: `if winv < qmax { to_be_processed.push(..) }`
                    } else {
                        alive[j] = false;
                    }
                }
                qb = wave_qmax(hmin, a.m);
            }
        }
        // ---- rounds i >= 2, all keys through a round before the next (pmh3a_more_points without the replay) ----
        for (uint32_t i = 2;; i++) {
            bool any = false;
#pragma unroll
            for (int j = 0; j <= NJ; j++) {
                if (__any(alive[j])) {
                    any = true;
                    if (alive[j]) {
                        const double qmax = __longlong_as_double((long long) qb);
                        const double hbase = winv[j] * (double) (i - 1);
                        if (!(hbase < qmax)) {
                            alive[j] = false;
                        } else {
                            const double x = exp01_sample(a.e01, rng[j]);
                            const double h = hbase + winv[j] * x;
                            const uint32_t k = draw_slot(a, rng[j]); // rounds >= 2 always draw the slot
                            if (h < qmax) slot_update_wave(hmin, sig, k, h, key[j]);
                            if (!(winv[j] * (double) i < qmax)) alive[j] = false;
                        }
                    }
                }
            }
            if (!any) break;
            qb = wave_qmax(hmin, a.m);
        }
        // ---- signature row: arg-min key per slot, initobj (0) for an empty multiset ----
        for (int t = lane; t < a.m; t += 64) {
            const uint64_t v = hmin[t] == H_INIT ? 0ull : sig[t];
            if (sig32) reinterpret_cast<uint32_t *>(a.sig_out)[(uint64_t) r * a.m + t] = (uint32_t) v;
            else reinterpret_cast<uint64_t *>(a.sig_out)[(uint64_t) r * a.m + t] = v;
        }
    }
}

// ---- k <= 8: the multiset of a read as a DIRECT-INDEXED histogram ---------------------------------------------------------
// Config 3 as the README times it (`datasketcher -k 8 -s 200`, src/bin/datasketcher.rs:222-254): 4^8 = 65 536 possible
// 8-mers, so the FnvHashMap<Kmer32bit::Val, u64> of seqsketchjaccard.rs:226-234 is an array in LDS indexed by the k-mer
// value itself (the canonical one for the canonical closures): one ds_add per position and nothing else -- no bucket ranks,
// no scan, no placement, no duplicate walk.  Counters are 16 bits wide, two to an LDS word, while the read has at most
// 65 535 k-mers (no counter can overflow); longer reads count with 32-bit counters in two passes over their positions, one per
// half of the index space.  The distinct k-mers are then enumerated -- from a list of first touches for reads of up to
// the list's capacity in k-mers, by scanning the histogram for longer ones -- and their ProbMinHash points are generated by the same
// code as everywhere else (pmh3a_first_point / pmh3a_more_points).  One persistent workgroup per CU, one read at a time.

__device__ __forceinline__ uint32_t revcomp32(uint32_t val, int k) {
    uint32_t rc = __brev(~val);
    rc = ((rc & 0x55555555u) << 1) | ((rc & 0xAAAAAAAAu) >> 1);
    return rc >> (32 - 2 * k);
}

// EMIT: the (key, weight) pairs of the distinct k-mers leave for k_pmh_points (one wave per read, 95 % VALU busy) instead of
// being turned into points here by a workgroup that has to meet at barriers.
template <bool EMIT>
__global__ void __launch_bounds__(1024) k_sketch_smallk(SketchArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem);
    uint64_t *hmin = reinterpret_cast<uint64_t *>(cnt + SMALLK_WORDS);
    uint64_t *sig = hmin + a.m;
    uint16_t *list = EMIT ? reinterpret_cast<uint16_t *>(hmin) : reinterpret_cast<uint16_t *>(sig + a.m); // (EMIT keeps no slots: the room goes to the list)
    const uint32_t list_cap = a.cap;                                         // entries, a multiple of 2048
    uint32_t *words = reinterpret_cast<uint32_t *>(list + list_cap);
    uint32_t *misc = words + SMALLK_TILE + 2; // [0] list length, [2..3] q_max, [4] first read, [5] the read after the current one
    uint64_t *qmax_sh = reinterpret_cast<uint64_t *>(misc + 2);
    const KmerCfg cfg = a.cfg;
    const int k = cfg.k, tid = threadIdx.x, nthreads = blockDim.x, wave = tid >> 6, nwaves = nthreads >> 6;
    const bool sig32 = a.sig_bytes == 4;
    const bool canonical = cfg.fhash == KMU_FHASH_CANON_RAW || cfg.fhash == KMU_FHASH_CANON_VALUE || cfg.fhash == KMU_FHASH_CANON_INVHASH;
    for (uint32_t i = tid; i < SMALLK_WORDS; i += nthreads) cnt[i] = 0;
    if (!EMIT)
        for (int t = tid; t < a.m; t += nthreads) { hmin[t] = H_INIT; sig[t] = 0; }
    // Reads are taken from the queue QCHUNK at a time (thread 0 keeps the cursor; the next chunk is requested one read before
    // it is needed).  While a read is being counted the NEXT read's header is fetched, and its first tile of code words is
    // requested right after the count phase: a fresh read starts without waiting for HBM.
    uint32_t q_next = 0, q_end = 0, q_pend = 0;
    bool q_pending = false;
    if (tid == 0) {
        misc[0] = 0; misc[1] = 0; *qmax_sh = H_INIT;
        q_next = atomicAdd(a.queue, (uint32_t) QCHUNK);
        q_end = q_next + QCHUNK;
        misc[4] = q_next++;
        misc[5] = q_next++; // (QCHUNK >= 2: the read after the first comes from the same chunk)
        if (q_next == q_end) { q_pend = atomicAdd(a.queue, (uint32_t) QCHUNK); q_pending = true; }
    }
    static_assert(QCHUNK >= 2, "the first two reads of a workgroup come from one chunk");
    lds_barrier();
    // thread 0: the read after the next one goes to misc[5] (called once per turn, after everybody has read misc[5] and
    // before the last barrier of the turn)
    auto q_post_next = [&]() {
        if (tid == 0) {
            if (q_next == q_end) {
                if (!q_pending) q_pend = atomicAdd(a.queue, (uint32_t) QCHUNK);
                q_next = q_pend;
                q_end = q_pend + QCHUNK;
                q_pending = false;
            }
            misc[5] = q_next++;
            if (q_next == q_end && !q_pending) { q_pend = atomicAdd(a.queue, (uint32_t) QCHUNK); q_pending = true; }
        }
    };
    // the key of histogram index `idx` (a k-mer value): the closure on that k-mer
    // (the README's closure -- canonical Kmer32bit through int32_hash, datasketcher.rs:225 -- without the walk through apply_fhash's cases)
    const bool fast32 = cfg.fhash == KMU_FHASH_CANON_INVHASH && cfg.kmer_type == KMU_KMER32BIT;
    auto key_of = [&](uint32_t idx) -> uint64_t {
        const uint32_t rc = revcomp32(idx, k);
        if (fast32) return (uint64_t) int32_hash((rc < idx ? rc : idx) | ((uint32_t) k << 28));
        return apply_fhash(cfg, (uint64_t) idx, (uint64_t) rc);
    };
    auto view_of = [&](uint32_t q) {
        SeqView v;
        v.base = a.bases;
        v.len = uniform_u64(a.offsets[q + 1] - a.offsets[q]);
        v.packed = a.packed;
        if (a.packed) {
            v.begin = uniform_u64(a.packed_offsets[q]);
            v.total = a.total_bytes ? a.total_bytes : uniform_u64(a.packed_offsets[a.n_seq - 1] + (a.offsets[a.n_seq] - a.offsets[a.n_seq - 1] + 3) / 4);
        } else {
            v.begin = uniform_u64(a.offsets[q]);
            v.total = a.total_bytes ? a.total_bytes : uniform_u64(a.offsets[a.n_seq]);
        }
        return v;
    };
    // words [0, nw] of a read's first tile (positions from 0): nw as the count loop computes it
    auto first_tile_nw = [&](const SeqView &v) -> uint32_t {
        const uint32_t Lv = v.len >= 0x80000000ull ? 0u : (uint32_t) v.len;
        if (Lv < (uint32_t) k) return 0u;
        const uint32_t nkv = Lv - (uint32_t) k + 1u, tpos = (SMALLK_TILE - 2) * 16, ld = seq_lead(v);
        const uint32_t t1 = nkv < tpos ? nkv : tpos;
        return (uint32_t) (((uint64_t) t1 - 1 + ld + (uint64_t) k - 1) >> 4) - (ld >> 4) + 1;
    };
    // (every barrier of this kernel orders LDS traffic only: the list stores and the prefetched loads stay in flight across it)
    uint32_t r = uniform_u32(misc[4]);
    SeqView sv = view_of(r < a.n_queue ? r : 0u);
    const uint64_t off_first = uniform_u64(a.offsets[0]);
    uint64_t off_r = uniform_u64(a.offsets[r < a.n_queue ? r : 0u]); // offsets[r] of the current read
    uint32_t pf_bad = 0; // non-ACGT bits of the words this thread staged ahead for the current read
    bool words_staged = false; // uniform: the current read's first tile was staged in LDS during the previous turn
    while (r < a.n_queue) {
        // (no barrier at the top of a turn: the last barrier of the previous turn made misc[5] and the pre-staged words visible)
        const uint32_t r_next = uniform_u32(misc[5]);
        // the next read's header: requested now, first looked at after the count phase (no wait here)
        const bool has_next = r_next < a.n_queue;
        const uint64_t n_o0 = has_next ? a.offsets[r_next] : 0ull, n_o1 = has_next ? a.offsets[r_next + 1] : 0ull;
        const uint64_t n_po = has_next && a.packed ? a.packed_offsets[r_next] : 0ull;
        SeqView nv = sv;
        bool nv_done = false; // uniform
        auto make_nv = [&]() {
            nv.len = uniform_u64(n_o1 - n_o0);
            nv.begin = a.packed ? uniform_u64(n_po) : uniform_u64(n_o0);
            nv_done = true;
        };
        if (sv.len >= 0x80000000ull && tid == 0) atomicOr(a.err, DERR_TABLE_FULL);
        const uint32_t L = sv.len >= 0x80000000ull ? 0u : (uint32_t) sv.len;
        const uint32_t nk = L >= (uint32_t) k ? L - (uint32_t) k + 1u : 0u;
        if (L == 0 && tid == 0) atomicOr(a.err, DERR_EMPTY_SEQ);
        uint32_t bad = words_staged ? pf_bad : 0u;
        if (nk == 0) bad |= wave_validate_seq(sv, wave, nwaves, false);
        const uint32_t lead = seq_lead(sv);
        const bool wide = nk > 65535u;         // 32-bit counters, two halves of the index space
        const bool listed = nk <= list_cap;    // first touches are listed: no histogram scan
        const uint32_t halves = wide ? 2u : 1u;
        const uint32_t tile_pos = (SMALLK_TILE - 2) * 16;
        uint32_t emit_n = 0; // EMIT: list entries of this read so far (uniform)
        for (uint32_t half = 0; half < halves && nk; half++) {
            // ---- count ----
            for (uint32_t tp0 = 0; tp0 < nk; tp0 += tile_pos) {
                const uint32_t tp1 = nk - tp0 > tile_pos ? tp0 + tile_pos : nk;
                const uint32_t wfirst = (tp0 + lead) >> 4;
                const uint32_t wlast = (uint32_t) (((uint64_t) tp1 - 1 + lead + (uint64_t) k - 1) >> 4);
                const uint32_t nw = wlast - wfirst + 1;
                if (words_staged && tp0 == 0 && half == 0) {
                    // the words were fetched and put into LDS while the previous read was handed over: its last barrier has
                    // made them visible -- a read's first tile starts counting at once
                } else {
                    if (tp0 != 0 || half != 0) lds_barrier(); // the previous tile's readers are done
                    for (uint32_t t = tid; t <= nw; t += nthreads) { // (+1: the word after the last, read by the window below)
                        uint32_t b;
                        words[t] = load_code_word(sv, (uint64_t) wfirst + t, b);
                        if (half == 0) bad |= b;
                    }
                    lds_barrier();
                }
                // four positions per thread and step: the window reads, then the four counter atomics, are requested
                // together, and the first touches of all four are appended with ONE atomic per wave (the loop is bound by
                // dependent LDS round trips, not by instructions)
                for (uint32_t p0 = tp0; p0 < tp1 && !ABL(1u); p0 += 4u * nthreads) { // uniform trip count (ballots)
                    uint32_t idx[4], old[4];
                    bool act[4], first[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t p = p0 + (uint32_t) u * nthreads + tid;
                        act[u] = p < tp1;
                        const uint32_t qq = (act[u] ? p : tp0) + lead - 16u * wfirst, wi = qq >> 4, sh = (qq & 15u) * 2u;
                        const uint64_t win = ((uint64_t) words[wi] << 32) | words[wi + 1];
                        const uint32_t val = (uint32_t) ((win << sh) >> (64 - 2 * k));
                        idx[u] = val;
                        if (canonical) { const uint32_t rc = revcomp32(val, k); idx[u] = rc < val ? rc : val; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        old[u] = 0;
                        if (!wide) { if (act[u]) old[u] = atomicAdd(&cnt[idx[u] >> 1], (idx[u] & 1u) ? 65536u : 1u); }
                        else if (act[u] && (idx[u] >> 15) == half) atomicAdd(&cnt[idx[u] & 0x7FFFu], 1u);
                    }
                    if (listed) {
                        uint32_t mine = 0, tot = 0, before = 0;
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            first[u] = act[u] && ((idx[u] & 1u) ? old[u] >> 16 : old[u] & 0xFFFFu) == 0u;
                            const uint64_t fm = __ballot(first[u]);
                            if (first[u]) mine = mine | (1u << u);
                            // rank of this lane's entry u among the wave's appends of this step: entries of earlier u first
                            if (first[u]) before = (before & ~(0xFFu << (8 * u))) | ((tot + (uint32_t) __popcll(fm & ((1ull << lane_id()) - 1ull))) << (8 * u));
                            tot += (uint32_t) __popcll(fm);
                        }
                        if (tot) { // wave-uniform
                            uint32_t base = 0;
                            if (lane_id() == 0) base = atomicAdd(&misc[0], tot);
                            base = bcast_u32(base, 0);
#pragma unroll
                            for (int u = 0; u < 4; u++)
                                if (mine & (1u << u)) list[base + ((before >> (8 * u)) & 0xFFu)] = (uint16_t) idx[u];
                        }
                    }
                }
            }
            lds_barrier();
            if (half + 1 == halves) {
                q_post_next(); // (everybody has read misc[5]; a barrier follows on every path below)
                if (has_next) { // the next read's first tile: fetched now and put into LDS (the staged words are free: this
                                // read is counted), the last barrier of this turn hands them to the next
                    make_nv();
                    const uint32_t nwn = first_tile_nw(nv), wf = seq_lead(nv) >> 4;
                    uint32_t b0 = 0, b1 = 0;
                    if (nwn) {
                        if ((uint32_t) tid <= nwn) words[tid] = load_code_word(nv, (uint64_t) wf + tid, b0);
                        if ((uint32_t) tid + nthreads <= nwn) words[tid + nthreads] = load_code_word(nv, (uint64_t) wf + tid + nthreads, b1);
                    }
                    pf_bad = b0 | b1;
                }
            }
            // ---- enumerate the distinct k-mers of this half, a list's worth at a time; pass 0 = first points, pass 1 = the
            //      further points of the keys whose 1 / w lies below the settled q_max (and the counters are wiped) ----
            const uint32_t wpb = wide ? list_cap : list_cap / 2; // histogram words whose counters fit the list
            const uint32_t nblocks = listed ? 1u : (SMALLK_WORDS + wpb - 1) / wpb;
            if (EMIT && listed && !wide) {
                // The common case in one sweep behind the count's barrier: every thread hands its first-touch entries over
                // and clears the counter it has just read (the two 16-bit counters of a word belong to different entries:
                // an atomic AND on the own half), one barrier, the list's length is reset -- the barrier at the end of the
                // read's turn orders that before the next read's appends.  Three barriers less per read than the general form.
                const uint32_t n_list = uniform_u32(misc[0]);
                const uint64_t lbase = off_r - off_first;
                for (uint32_t i = tid; i < n_list && !ABL(2u); i += nthreads) {
                    const uint32_t idx = list[i];
                    const uint32_t c = cnt[idx >> 1];
                    a.lst_keys[lbase + i] = key_of(idx);
                    a.lst_w[lbase + i] = (idx & 1u) ? c >> 16 : c & 0xFFFFu;
                    atomicAnd(&cnt[idx >> 1], (idx & 1u) ? 0x0000FFFFu : 0xFFFF0000u);
                }
                emit_n = n_list;
                lds_barrier();
                if (tid == 0) misc[0] = 0;
                continue;
            }
            for (int pass = EMIT ? 1 : 0; pass < 2; pass++) { // (EMIT: one traversal: hand over, wipe)
                uint64_t qb = (!EMIT && pass) ? wave_qmax(hmin, a.m) : 0ull;
                for (uint32_t blk = 0; blk < nblocks; blk++) {
                    if (!listed) { // the occupied counters of this block of the histogram
                        if (tid == 0) misc[0] = 0;
                        lds_barrier();
                        const uint32_t w0 = blk * wpb, w1 = w0 + wpb < SMALLK_WORDS ? w0 + wpb : SMALLK_WORDS;
                        for (uint32_t wbase = w0; wbase < w1; wbase += nthreads) { // uniform trip count (ballots)
                            const uint32_t c = wbase + tid < w1 ? cnt[wbase + tid] : 0u;
                            for (uint32_t h = 0; h < (wide ? 1u : 2u); h++) {
                                const bool occ = wide ? c != 0u : ((h ? c >> 16 : c & 0xFFFFu) != 0u);
                                const uint64_t om = __ballot(occ);
                                if (om) {
                                    const int leader = __ffsll((unsigned long long) om) - 1;
                                    uint32_t base = 0;
                                    if (lane_id() == leader) base = atomicAdd(&misc[0], (uint32_t) __popcll(om));
                                    base = bcast_u32(base, leader);
                                    if (occ) {
                                        const uint32_t idx = wide ? ((half << 15) | (wbase + tid)) : (((wbase + tid) << 1) | h);
                                        list[base + (uint32_t) __popcll(om & ((1ull << lane_id()) - 1ull))] = (uint16_t) idx;
                                    }
                                }
                            }
                        }
                    }
                    lds_barrier();
                    const uint32_t n_list = uniform_u32(misc[0]);
                    uint32_t chunk = 0;
                    if (EMIT) {
                        const uint64_t lbase = off_r - off_first + emit_n;
                        for (uint32_t i = tid; i < n_list && !ABL(2u); i += nthreads) {
                            const uint32_t idx = list[i];
                            const uint32_t c = wide ? cnt[idx & 0x7FFFu] : cnt[idx >> 1];
                            a.lst_keys[lbase + i] = key_of(idx);
                            a.lst_w[lbase + i] = wide ? c : ((idx & 1u) ? c >> 16 : c & 0xFFFFu);
                        }
                        emit_n += n_list;
                    } else
                    for (uint32_t base = 0; base < n_list; base += nthreads, chunk++) {
                        const uint32_t i = base + tid;
                        uint32_t idx = 0, w = 0;
                        if (i < n_list) {
                            idx = list[i];
                            const uint32_t c = wide ? cnt[idx & 0x7FFFu] : cnt[idx >> 1];
                            w = wide ? c : ((idx & 1u) ? c >> 16 : c & 0xFFFFu);
                        }
                        const bool have = w != 0u;
                        if (pass == 0) {
                            if (__any(have))
                                (void) pmh3a_first_point(a, sig32, hmin, sig, qmax_sh, ((chunk + wave) & B1_REFRESH_MASK) == 0u, have,
                                                         have ? key_of(idx) : 0ull, w);
                        } else {
                            double winv = 0.0;
                            bool alive = false;
                            if (have) {
                                winv = 1.0 / (double) w;
                                alive = winv < __longlong_as_double((long long) qb);
                            }
                            if (__any(alive)) pmh3a_more_points(a, sig32, hmin, sig, qb, alive, alive ? key_of(idx) : 0ull, winv);
                        }
                    }
                    lds_barrier(); // every reader of the list and of the counters is through
                    if (pass == 1) { // wipe what this block enumerated
                        if (listed) {
                            for (uint32_t i = tid; i < n_list; i += nthreads) cnt[wide ? (list[i] & 0x7FFFu) : (list[i] >> 1)] = 0u;
                        } else {
                            const uint32_t w0 = blk * wpb, w1 = w0 + wpb < SMALLK_WORDS ? w0 + wpb : SMALLK_WORDS;
                            for (uint32_t i = w0 + tid; i < w1; i += nthreads) cnt[i] = 0u;
                        }
                    }
                }
                lds_barrier(); // (pass 0 -> 1: all first points are in before q_max is read)
            }
            if (tid == 0) misc[0] = 0;
            lds_barrier();
        }
        if (bad) atomicOr(a.err, DERR_NON_ACGT);
        // ---- signature row: arg-min key per slot, initobj (0) for an empty multiset ----
        if (EMIT) {
            if (tid == 0) a.lst_n[r] = emit_n; // the row is written by k_pmh_points
        } else
        for (int t = tid; t < a.m; t += nthreads) {
            const uint64_t v = hmin[t] == H_INIT ? 0ull : sig[t];
            if (sig32) reinterpret_cast<uint32_t *>(a.sig_out)[(uint64_t) r * a.m + t] = (uint32_t) v;
            else reinterpret_cast<uint64_t *>(a.sig_out)[(uint64_t) r * a.m + t] = v;
            hmin[t] = H_INIT;
            sig[t] = 0;
        }
        if (tid == 0) *qmax_sh = H_INIT;
        // (a read without k-mers never reached the prefetch: its successor loads its own words; its turn has no barrier of
        //  its own either, so the queue is advanced behind one here)
        if (nk == 0) {
            q_post_next();
            lds_barrier();
        }
        if (has_next && !nv_done) make_nv();
        words_staged = nk != 0 && has_next && first_tile_nw(nv) != 0;
        r = r_next;
        sv = nv;
        off_r = uniform_u64(n_o0);
    }
}

// merge the slot minima of disjoint key sets (leaves): per slot the smallest (h, key); one workgroup per slot
// (stride: words between the rows of consecutive parts; part_out: write (h, key) to part_out[t], part_out[m + t] instead)
__global__ void __launch_bounds__(256) k_pmh_reduce(const uint64_t *part_h, const uint64_t *part_k, uint64_t n_parts, int m,
                                                    uint64_t stride, int sig_bytes, void *sig_out, uint64_t *part_out) {
    __shared__ uint64_t sh[256], sk[256];
    const int t = blockIdx.x;
    uint64_t bh = H_INIT, bk = 0;
    for (uint64_t i = threadIdx.x; i < n_parts; i += blockDim.x) {
        const uint64_t h = part_h[i * stride + t], key = part_k[i * stride + t];
        if (h < bh || (h == bh && h != H_INIT && key < bk)) { bh = h; bk = key; }
    }
    sh[threadIdx.x] = bh;
    sk[threadIdx.x] = bk;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int) threadIdx.x < d) {
            const uint64_t h = sh[threadIdx.x + d], key = sk[threadIdx.x + d];
            if (h < sh[threadIdx.x] || (h == sh[threadIdx.x] && h != H_INIT && key < sk[threadIdx.x])) {
                sh[threadIdx.x] = h;
                sk[threadIdx.x] = key;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (part_out) {
            part_out[t] = sh[0];
            part_out[m + t] = sk[0];
            return;
        }
        const uint64_t v = sh[0] == H_INIT ? 0ull : sk[0];
        if (sig_bytes == 4) reinterpret_cast<uint32_t *>(sig_out)[t] = (uint32_t) v;
        else reinterpret_cast<uint64_t *>(sig_out)[t] = v;
    }
}

// longest sequence: out[0] = max_i (offsets[i + 1] - offsets[i]); out[0] must be 0 on entry.  out[1] = sum of the lengths
__global__ void __launch_bounds__(1024) k_max_len(const uint64_t *offsets, uint32_t n_seq, uint64_t *out) {
    uint64_t mx = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_seq; i += gridDim.x * blockDim.x) {
        const uint64_t L = offsets[i + 1] - offsets[i];
        mx = L > mx ? L : mx;
    }
    mx = wave_max_u64(mx);
    if (lane_id() == 0 && mx) atomicMax((unsigned long long *) out, (unsigned long long) mx);
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = offsets[n_seq] - offsets[0]; // all bases of the call
}

// exclusive scan of the k-mer counts max(0, L_i - k + 1) of all sequences (single workgroup); koff[n] = total
__global__ void __launch_bounds__(1024) k_nk_scan(const uint64_t *offsets, uint32_t n_seq, int k, uint64_t *koff,
                                                  uint32_t *err) {
    __shared__ uint64_t wtot[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_seq; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        uint64_t v = 0;
        if (i < n_seq) {
            const uint64_t L = offsets[i + 1] - offsets[i];
            if (L == 0) atomicOr(err, 8u);
            v = L >= (uint64_t) k ? L - k + 1 : 0;
        }
        uint64_t incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t o = ((uint64_t) (uint32_t) __shfl_up((int) (incl >> 32), d, 64) << 32) |
                         (uint32_t) __shfl_up((int) (uint32_t) incl, d, 64);
            if (lane_id() >= d) incl += o;
        }
        if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint64_t pre = carry;
        for (int w = 0; w < (int) (threadIdx.x >> 6); w++) pre += wtot[w];
        if (i < n_seq) koff[i] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry = pre + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) koff[n_seq] = carry;
}

// fhash(kmer) of every k-mer of every sequence, compact: out[koff[i] + p]
__global__ void __launch_bounds__(256) k_seq_hashes_compact(const uint8_t *bases, const uint64_t *offsets,
                                                            const uint64_t *packed_offsets, uint32_t n_seq, int packed,
                                                            uint64_t total, KmerCfg cfg, const uint64_t *koff, uint64_t *out,
                                                            uint32_t *err, int spread) {
    // spread = 0: one workgroup per sequence (many sequences); spread = 1: every sequence is walked by the whole grid
    // (a few long sequences, e.g. the contigs of a genome)
    const int wave = spread ? (int) (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) : (int) (threadIdx.x >> 6);
    const int nwaves = spread ? (int) (gridDim.x * (blockDim.x >> 6)) : (int) (blockDim.x >> 6);
    const bool aa = cfg.kmer_type == KMU_KMERAA32BIT || cfg.kmer_type == KMU_KMERAA64BIT;
    for (uint32_t i = spread ? 0u : blockIdx.x; i < n_seq; i += spread ? 1u : gridDim.x) {
        SeqView s;
        s.base = bases;
        s.len = offsets[i + 1] - offsets[i];
        s.packed = packed;
        if (packed) {
            s.begin = packed_offsets[i];
            s.total = total ? total : (packed_offsets[n_seq - 1] + (offsets[n_seq] - offsets[n_seq - 1] + 3) / 4);
        } else {
            s.begin = offsets[i];
            s.total = total ? total : offsets[n_seq];
        }
        const uint64_t nk = s.len >= (uint64_t) cfg.k ? s.len - cfg.k + 1 : 0;
        uint64_t *o = out + koff[i];
        uint32_t bad = 0;
        if (nk == 0) bad |= wave_validate_seq(s, wave, nwaves, aa);
        else if (aa) {
            for (uint64_t st = wave; st < (s.len + 63) / 64; st += nwaves)
                bad |= wave_step_kmers_aa(s, cfg.k, st, 0, nk, [&](uint64_t p, uint64_t val, uint64_t) { o[p] = apply_fhash(cfg, val, 0); });
        } else {
            for (uint64_t st = wave; st < (seq_num_words(s) + 63) / 64; st += nwaves)
                bad |= wave_step_kmers(s, cfg.k, st, 0, nk, [&](uint64_t p, uint64_t val, uint64_t rc) { o[p] = apply_fhash(cfg, val, rc); });
        }
        if (bad) atomicOr(err, aa ? DERR_BAD_AA : DERR_NON_ACGT);
    }
}

__global__ void __launch_bounds__(256) k_widen_u32(const uint32_t *in, uint64_t n, uint64_t *out) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = in[i];
}

// the forms the host side launches (kmu_sketch_kernels.h)
#define KMU_X_INST(...) template __global__ void __VA_ARGS__(SketchArgs);
KMU_SKETCH_KERNEL_FORMS(KMU_X_INST)
#undef KMU_X_INST

} // namespace kmu
