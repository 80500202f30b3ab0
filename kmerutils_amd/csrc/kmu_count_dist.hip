// kmu_count_dist.hip -- the distributed counter (KMU_COUNT_DISTRIBUTED; kmu.h "multi-GPU"): one member of a KmerCounterPool
// (src/base/kmercount.rs:424-565) spread over the ranks of a communicator.  After kmu_count_finalize rank r holds exactly the
// canonical k-mers it owns with their multiplicities over all ranks' reads.  The owner functions are in kmu_count_table.h /
// kmu_smer.h, the grouping kernels in kmu_count_part.hip (hash owners) and kmu_smer.hip (minimizer owners).
#include <algorithm>
#include <vector>

#include "kmu_count_table.h"

namespace kmu {


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
int dist_add_begin(kmu_counter *c, DevSeqs &ds, uint64_t total_bases, uint32_t *d_err) {
    kmu_ctx *ctx = c->ctx;
    kmu_comm *cm = ctx->comm;
    if (!cm) return fail(ctx, KMU_E_BAD_ARG, "the communicator of this distributed counter's context is gone (kmu_comm_destroy)");
    const uint32_t N = (uint32_t) cm->nranks;
    const bool smer = c->okind == 1;
    comm_stats_reset(ctx);
    cm->stats.owner_kind = c->okind;
    // ---- census + sample ----
    void *slist, *sn;
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
    if (n_s && !h_sn[1]) KMU_TRY(sample_distinct(ctx, (const uint64_t *) slist, n_s, (uint32_t *) sn + 2, &d_s));
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
    // the table of a counter created with KMU_COUNT_HINT_OCCURRENCES: a rank ends up owning what the owner function gives it -- about
    // 1 / N of the job's k-mers with hash owners, more or less than that with minimizer owners on repetitive data -- so the size
    // is taken from the gathered per-owner counts (the most any owner receives; the MERGE route holds its shard's own first: no
    // more than a shard's worth either); the same size on every rank
    double own_max = 0;
    for (uint32_t p = 0; p < N; p++) {
        double s = 0;
        for (uint32_t r = 0; r < N; r++) s += (double) all[(size_t) r * RW + H + N + p];
        own_max = std::max(own_max, s);
    }
    KMU_TRY(table_alloc_for(c, nullptr, 0, d_err, ratio > 0 ? ratio : 1.0, (uint64_t) std::max(n_loc, own_max) + 1));
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
int dist_add_end(kmu_counter *c) {
    kmu_ctx *ctx = c->ctx;
    if (!c->pending) return KMU_OK;
    c->pending = false;
    if (!ctx->comm) return fail(ctx, KMU_E_BAD_ARG, "the communicator went away under an exchange of this counter (kmu_comm_destroy)");
    KMU_TRY(comm_wait(ctx));
    if (c->pend_recv == 0) return KMU_OK;
    if (c->okind == 1) return add_superkmers(c, ctx->bufs["cnt.recv"].p, c->pend_recv, c->pend_kmers);
    return add_entries(c, (const uint64_t *) ctx->bufs["cnt.recv"].p, nullptr, c->pend_recv, KMU_MEM_DEVICE);
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

} // namespace kmu

using namespace kmu;

extern "C" {

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
    KMU_TRY(comm_wait(ctx));
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

} // extern "C"
