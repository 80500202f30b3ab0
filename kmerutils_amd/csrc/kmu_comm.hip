// kmu_comm.hip -- communicator of a context: RCCL over xGMI (one rank per GPU), or a transport supplied by the host.
//
// Reference shape: the reference's "ranks" are threads of one process that talk through crossbeam channels
// (count_kmer_threaded_one_to_many, src/base/kmercount.rs:881-974: one producer dispatches every canonical k-mer to the
// thread that owns it; KmerCounterPool :424-565 keeps one counter per thread).  Here a rank is a GPU, the channel is one
// all-to-all over xGMI, and the calls below are what a Rust host binds: no Python, no torch.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <mutex>

#include "kmu_comm.hpp"

namespace kmu {

static_assert(sizeof(ncclUniqueId) == KMU_COMM_ID_BYTES, "kmu_comm_id carries an ncclUniqueId");

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// One RCCL per process: the copy that is already mapped (RTLD_NOLOAD; torch's has the SONAME librccl.so.1 too), else the
// system's.
static Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        const char *paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : paths)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) {
            const char *e = dlerror();
            r.error = std::string("cannot load librccl.so.1: ") + (e ? e : "?");
            return;
        }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId)) sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank)) sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy)) sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather)) sym("ncclAllGather");
        r.Send = (decltype(r.Send)) sym("ncclSend");
        r.Recv = (decltype(r.Recv)) sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart)) sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd)) sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString)) sym("ncclGetErrorString");
    });
    return &r;
}

#define KMU_NCCL(ctx, expr)                                                                                         \
    do {                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                   \
        if (r_ != ncclSuccess)                                                                                      \
            return kmu::fail((ctx), KMU_E_RCCL, "%s: %s (%s:%d)", #expr, kmu::rccl()->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

int comm_allgather_host(kmu_ctx *ctx, const void *send, void *recv, uint64_t bytes) {
    kmu_comm *c = ctx->comm;
    if (!c) return fail(ctx, KMU_E_BAD_ARG, "the context has no communicator (kmu_comm_init)");
    if (c->ag) {
        const int rc = c->ag(c->user, send, recv, bytes);
        if (rc) return fail(ctx, KMU_E_RCCL, "the host's all-gather failed (%d)", rc);
        return KMU_OK;
    }
    void *d;
    KMU_TRY(dev_buf(ctx, "comm.meta", (size_t) bytes * ((size_t) c->nranks + 1) + 64, &d));
    uint8_t *d_send = (uint8_t *) d, *d_recv = d_send + ((bytes + 15) & ~(uint64_t) 15);
    KMU_HIP(ctx, hipMemcpyAsync(d_send, send, bytes, hipMemcpyHostToDevice, ctx->stream));
    KMU_NCCL(ctx, rccl()->AllGather(d_send, d_recv, bytes, ncclUint8, (ncclComm_t) c->nccl, ctx->stream));
    KMU_HIP(ctx, hipMemcpyAsync(recv, d_recv, bytes * (uint64_t) c->nranks, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KMU_OK;
}

// The COPY transport's all-to-all (kmu_comm::copy).  Every rank publishes where its receive buffer is -- an IPC handle of the
// allocation, its process id and address for peers that are threads of the same process -- and where every peer's share starts
// in it; the all-gather that carries this is also the point at which every rank's earlier work on both buffers is known to be
// finished (each rank drains its stream before it publishes: a peer writes into memory this rank may still have been reading).
// Then N - 1 device-to-device copies on the exchange stream (+ the rank's own share), closed by comm_wait's barrier.
#include <unistd.h>
static int carrier_alltoallv(kmu_ctx *ctx, const void *send_dev, const uint64_t *send_counts, const uint64_t *send_displs, void *recv_dev,
                             const uint64_t *recv_counts, const uint64_t *recv_displs, uint32_t elem_bytes, hipStream_t s);
static int copy_alltoallv(kmu_ctx *ctx, const void *send_dev, const uint64_t *send_counts, const uint64_t *send_displs, void *recv_dev,
                          const uint64_t *recv_counts, const uint64_t *recv_displs, uint32_t elem_bytes, hipStream_t s) {
    kmu_comm *c = ctx->comm;
    const int N = c->nranks;
    if (c->copy_pending) KMU_TRY(comm_wait(ctx)); // (an exchange nobody waited for: its barrier first)
    c->peers.resize((size_t) N);
    uint64_t n_in = 0;
    for (int p = 0; p < N; p++)
        if (p != c->rank) n_in += recv_counts[p];
    // row of a rank: [0] pid, [1] address of its receive buffer, [2] 1 = the handle is valid, [3 .. 3 + N) where peer p's share starts
    // in it (bytes), then the 64 bytes of the handle
    const size_t row_words = 3 + (size_t) N + sizeof(hipIpcMemHandle_t) / 8;
    static_assert(sizeof(hipIpcMemHandle_t) % 8 == 0, "the handle travels as 64-bit words");
    std::vector<uint64_t> mine(row_words, 0), all(row_words * (size_t) N, 0);
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the send buffer is complete, the receive buffer no longer in use
    KMU_HIP(ctx, hipStreamSynchronize(s));
    mine[0] = (uint64_t) getpid();
    mine[1] = (uint64_t) (uintptr_t) recv_dev;
    if (n_in && recv_dev) {
        hipIpcMemHandle_t h;
        const char *nx = getenv("KMU_COMM_NO_EXPORT"); // tests: this rank's export "fails"
        if (!(nx && atoi(nx) == c->rank) && hipIpcGetMemHandle(&h, recv_dev) == hipSuccess) {
            mine[2] = 1;
            memcpy(&mine[3 + N], &h, sizeof h);
        } else {
            (void) hipGetLastError();
            mine[2] = 2; // wanted and not to be had (seen once in five runs of the two-process test, in the second exchange of a finalize)
        }
    }
    for (int p = 0; p < N; p++) mine[3 + p] = recv_displs[p] * elem_bytes;
    KMU_TRY(comm_allgather_host(ctx, mine.data(), all.data(), (uint64_t) row_words * 8));
    // A rank that could not export its buffer: every rank reads the same rows, so all of them take THIS exchange through the
    // communicator's other data path (the host's all-to-all or RCCL) where it has one -- unless every peer of that rank is a thread of
    // its process (its plain pointer will do).
    for (int q = 0; q < N; q++) {
        const uint64_t *row = &all[row_words * (size_t) q];
        if (row[2] != 2) continue;
        bool needed = false;
        for (int p = 0; p < N; p++) needed = needed || all[row_words * (size_t) p] != row[0];
        if (!needed) continue;
        if (c->a2a || c->nccl) {
            c->carrier_pending = true;
            return carrier_alltoallv(ctx, send_dev, send_counts, send_displs, recv_dev, recv_counts, recv_displs, elem_bytes, s);
        }
        return fail(ctx, KMU_E_RCCL, "rank %d could not export its receive buffer (hipIpcGetMemHandle) and the communicator has no other data path", q);
    }
    hipEvent_t t_a = nullptr, t_b = nullptr;
    auto take_event = [&]() -> hipEvent_t {
        hipEvent_t e = nullptr;
        if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
        else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        return e;
    };
    t_a = take_event();
    t_b = take_event();
    auto give_back = [&]() {
        if (t_a) c->ev_pool.push_back(t_a);
        if (t_b) c->ev_pool.push_back(t_b);
    };
    if (t_a) (void) hipEventRecord(t_a, s);
    const uint8_t *sb = (const uint8_t *) send_dev;
    for (int q = 0; q < N; q++) {
        const uint64_t bytes = send_counts[q] * elem_bytes;
        if (!bytes) continue;
        const uint64_t *row = &all[row_words * (size_t) q];
        uint8_t *dst = nullptr;
        if (q == c->rank) dst = (uint8_t *) recv_dev;
        else if (row[0] == mine[0]) dst = (uint8_t *) (uintptr_t) row[1]; // a thread of this process: its pointer is good here
        else {
            kmu_comm::Peer &pe = c->peers[(size_t) q];
            if (row[2] != 1) { give_back(); return fail(ctx, KMU_E_RCCL, "rank %d published no handle of its receive buffer", q); }
            if (!pe.opened || memcmp(&pe.handle, &row[3 + N], sizeof pe.handle) != 0) { // the peer's buffer is new (or has grown)
                if (pe.opened) (void) hipIpcCloseMemHandle(pe.base);
                pe.opened = false;
                memcpy(&pe.handle, &row[3 + N], sizeof pe.handle);
                const hipError_t e = hipIpcOpenMemHandle(&pe.base, pe.handle, hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) {
                    (void) hipGetLastError();
                    give_back();
                    return fail(ctx, KMU_E_RCCL, "hipIpcOpenMemHandle of rank %d's receive buffer: %s", q, hipGetErrorString(e));
                }
                pe.opened = true;
            }
            dst = (uint8_t *) pe.base;
        }
        const hipError_t e = hipMemcpyAsync(dst + row[3 + c->rank], sb + send_displs[q] * elem_bytes, (size_t) bytes, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) {
            (void) hipGetLastError();
            give_back();
            return fail(ctx, KMU_E_HIP, "the copy of rank %d's share of an exchange failed: %s", q, hipGetErrorString(e));
        }
    }
    if (t_a && t_b) {
        (void) hipEventRecord(t_b, s);
        c->timed.emplace_back(t_a, t_b);
    } else give_back();
    c->copy_pending = true;
    return KMU_OK;
}

int comm_wait(kmu_ctx *ctx) {
    kmu_comm *c = ctx->comm;
    if (!c) return fail(ctx, KMU_E_BAD_ARG, "the context has no communicator (kmu_comm_init)");
    if (c->copy) {
        if (c->carrier_pending) { // (an exchange of the COPY transport that went through the other data path)
            c->carrier_pending = false;
            if (!c->a2a) KMU_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_b, 0));
        }
        if (!c->copy_pending) return KMU_OK;
        c->copy_pending = false;
        KMU_HIP(ctx, hipStreamSynchronize(c->stream)); // this rank's copies have landed at its peers ...
        uint64_t one = 1;
        std::vector<uint64_t> all((size_t) c->nranks);
        return comm_allgather_host(ctx, &one, all.data(), 8); // ... and everybody else's here, once every rank has said so
    }
    if (!c->a2a) KMU_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_b, 0));
    return KMU_OK;
}

int comm_alltoallv(kmu_ctx *ctx, const void *send_dev, const uint64_t *send_counts, const uint64_t *send_displs, void *recv_dev,
                   const uint64_t *recv_counts, const uint64_t *recv_displs, uint32_t elem_bytes, hipStream_t s) {
    kmu_comm *c = ctx->comm;
    if (!c) return fail(ctx, KMU_E_BAD_ARG, "the context has no communicator (kmu_comm_init)");
    uint64_t out = 0, in = 0;
    for (int p = 0; p < c->nranks; p++)
        if (p != c->rank) { out += send_counts[p] * elem_bytes; in += recv_counts[p] * elem_bytes; }
    c->stats.bytes_sent += out;
    c->stats.bytes_received += in;
    c->stats.exchanges++;
    if (c->copy) return copy_alltoallv(ctx, send_dev, send_counts, send_displs, recv_dev, recv_counts, recv_displs, elem_bytes, s);
    return carrier_alltoallv(ctx, send_dev, send_counts, send_displs, recv_dev, recv_counts, recv_displs, elem_bytes, s);
}

// the exchange through the host's all-to-all (kmu_comm_init_custom) or RCCL
static int carrier_alltoallv(kmu_ctx *ctx, const void *send_dev, const uint64_t *send_counts, const uint64_t *send_displs, void *recv_dev,
                             const uint64_t *recv_counts, const uint64_t *recv_displs, uint32_t elem_bytes, hipStream_t s) {
    kmu_comm *c = ctx->comm;
    if (c->a2a) {
        KMU_HIP(ctx, hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = c->a2a(c->user, send_dev, send_counts, send_displs, recv_dev, recv_counts, recv_displs, elem_bytes, (void *) s);
        c->stats.exchange_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rc) return fail(ctx, KMU_E_RCCL, "the host's all-to-all failed (%d)", rc);
        return KMU_OK;
    }
    // the exchange is timed where it runs: an event pair around it on the exchange stream, read by kmu_comm_get_stats
    auto take_event = [&]() -> hipEvent_t {
        hipEvent_t e = nullptr;
        if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
        else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        return e;
    };
    hipEvent_t t_a = take_event(), t_b = take_event();
    auto give_back = [&]() { // (every way out that does not hand the pair to `timed`)
        if (t_a) c->ev_pool.push_back(t_a);
        if (t_b) c->ev_pool.push_back(t_b);
        t_a = t_b = nullptr;
    };
    if (t_a) (void) hipEventRecord(t_a, s);
    const uint8_t *sb = (const uint8_t *) send_dev;
    uint8_t *rb = (uint8_t *) recv_dev;
    // what a rank keeps does not travel: one device-to-device copy (no RCCL channel, no peer).  KMU_COMM_SELF_RCCL=1 sends it to
    // self through RCCL like a peer's share (tests and measurements on one GPU: the only RCCL data path there is)
    const char *sr = getenv("KMU_COMM_SELF_RCCL");
    const bool self_rccl = sr && atoi(sr) != 0;
    if (!self_rccl && send_counts[c->rank] && hipMemcpyAsync(rb + recv_displs[c->rank] * elem_bytes, sb + send_displs[c->rank] * elem_bytes,
                                               (size_t) send_counts[c->rank] * elem_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        give_back();
        return fail(ctx, KMU_E_HIP, "the copy of a rank's own share of an exchange failed: %s", hipGetErrorString(hipGetLastError()));
    }
    // every pair of ranks exchanges one message per round: xGMI is point to point, all seven links of a GPU carry traffic at
    // once.  A message is at most KMU_COMM_CHUNK_MB (default 1024) long: rounds of grouped sends / receives until every
    // pair is through (sizes are known on both sides, so both sides run the same number of rounds for a pair).
    uint64_t chunk = 1024ull << 20;
    if (const char *e = getenv("KMU_COMM_CHUNK_MB")) chunk = (uint64_t) std::max(1, atoi(e)) << 20;
    uint64_t longest = 0;
    for (int p = 0; p < c->nranks; p++)
        if (p != c->rank || self_rccl) longest = std::max(longest, std::max(send_counts[p], recv_counts[p]) * elem_bytes);
    for (uint64_t o = 0; o < longest; o += chunk) {
        const ncclResult_t gs = rccl()->GroupStart();
        if (gs != ncclSuccess) {
            give_back();
            return fail(ctx, KMU_E_RCCL, "ncclGroupStart: %s", rccl()->GetErrorString(gs));
        }
        // a failing send / receive must not leave the group open (every later RCCL call of this thread would queue into it and
        // the peers that have posted their side would wait): the first error is kept, the group is always closed
        ncclResult_t first = ncclSuccess;
        for (int p = 0; p < c->nranks && first == ncclSuccess; p++) {
            if (p == c->rank && !self_rccl) continue;
            const uint64_t sbytes = send_counts[p] * elem_bytes, rbytes = recv_counts[p] * elem_bytes;
            if (o < sbytes)
                first = rccl()->Send(sb + send_displs[p] * elem_bytes + o, (size_t) std::min(chunk, sbytes - o), ncclUint8, p, (ncclComm_t) c->nccl, s);
            if (first == ncclSuccess && o < rbytes)
                first = rccl()->Recv(rb + recv_displs[p] * elem_bytes + o, (size_t) std::min(chunk, rbytes - o), ncclUint8, p, (ncclComm_t) c->nccl, s);
        }
        const ncclResult_t ge = rccl()->GroupEnd();
        if (first != ncclSuccess || ge != ncclSuccess) {
            give_back();
            if (first != ncclSuccess) return fail(ctx, KMU_E_RCCL, "ncclSend / ncclRecv in the all-to-all: %s", rccl()->GetErrorString(first));
            return fail(ctx, KMU_E_RCCL, "ncclGroupEnd: %s", rccl()->GetErrorString(ge));
        }
    }
    if (t_a && t_b) {
        (void) hipEventRecord(t_b, s);
        c->timed.emplace_back(t_a, t_b);
    } else give_back();
    return KMU_OK;
}

// the event pairs of FINISHED exchanges -> stats.exchange_ms; one still in flight stays queued for the next query (a statistics
// call never stalls the host on the exchange stream)
static void comm_collect_timed(kmu_comm *c) {
    size_t kept = 0;
    for (auto &pr : c->timed) {
        if (hipEventQuery(pr.second) == hipErrorNotReady) { c->timed[kept++] = pr; continue; }
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) c->stats.exchange_ms += ms;
        c->ev_pool.push_back(pr.first);
        c->ev_pool.push_back(pr.second);
    }
    c->timed.resize(kept);
    (void) hipGetLastError();
}

void comm_stats_reset(kmu_ctx *ctx) {
    kmu_comm *c = ctx->comm;
    if (!c) return;
    for (auto &pr : c->timed) {
        c->ev_pool.push_back(pr.first);
        c->ev_pool.push_back(pr.second);
    }
    c->timed.clear();
    c->stats = kmu_comm_stats{};
}

static int comm_new(kmu_ctx *ctx, int rank, int nranks) {
    if (ctx->comm) return fail(ctx, KMU_E_BAD_ARG, "the context already has a communicator");
    if (nranks < 1 || nranks > 2048 || rank < 0 || rank >= nranks) return fail(ctx, KMU_E_BAD_ARG, "bad rank %d of %d", rank, nranks);
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    kmu_comm *c = new kmu_comm();
    c->rank = rank;
    c->nranks = nranks;
    // the exchange stream at the device's highest priority: its kernels (RCCL's copies, a blit of the rank's own share) compete for
    // CUs with the sketch kernels that run under the exchange -- at every kernel boundary of the other stream they go first
    int prio_lo = 0, prio_hi = 0;
    (void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi) != hipSuccess || hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return fail(ctx, KMU_E_HIP, "cannot create the exchange stream");
    }
    ctx->comm = c;
    return KMU_OK;
}

void comm_free(kmu_ctx *ctx) {
    kmu_comm *c = ctx->comm;
    if (!c) return;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    for (auto &pe : c->peers)
        if (pe.opened) (void) hipIpcCloseMemHandle(pe.base);
    if (c->nccl) (void) rccl()->CommDestroy((ncclComm_t) c->nccl);
    for (auto &pr : c->timed) { (void) hipEventDestroy(pr.first); (void) hipEventDestroy(pr.second); }
    for (hipEvent_t e : c->ev_pool) (void) hipEventDestroy(e);
    if (c->ev_a) (void) hipEventDestroy(c->ev_a);
    if (c->ev_b) (void) hipEventDestroy(c->ev_b);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
    ctx->comm = nullptr;
}

} // namespace kmu

using namespace kmu;

extern "C" {

int kmu_comm_get_id(kmu_comm_id *out) {
    if (!out) return KMU_E_BAD_ARG;
    Rccl *r = rccl();
    if (!r->error.empty()) return fail(nullptr, KMU_E_RCCL, "%s", r->error.c_str());
    ncclUniqueId id;
    const ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return fail(nullptr, KMU_E_RCCL, "ncclGetUniqueId: %s", r->GetErrorString(rc));
    memcpy(out->bytes, &id, KMU_COMM_ID_BYTES);
    return KMU_OK;
}

int kmu_comm_init(kmu_ctx *ctx, const kmu_comm_id *id, int rank, int nranks) {
    if (!ctx || !id) return KMU_E_BAD_ARG;
    Rccl *r = rccl();
    if (!r->error.empty()) return fail(ctx, KMU_E_RCCL, "%s", r->error.c_str());
    KMU_TRY(comm_new(ctx, rank, nranks));
    ncclUniqueId nid;
    memcpy(&nid, id->bytes, KMU_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r->CommInitRank(&comm, nranks, nid, rank);
    if (rc != ncclSuccess) {
        comm_free(ctx);
        return fail(ctx, KMU_E_RCCL, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, r->GetErrorString(rc));
    }
    ctx->comm->nccl = comm;
    return KMU_OK;
}

int kmu_comm_init_custom(kmu_ctx *ctx, int rank, int nranks, kmu_alltoallv_fn alltoallv, kmu_allgather_fn allgather, void *user) {
    if (!ctx || !allgather) return KMU_E_BAD_ARG;
    KMU_TRY(comm_new(ctx, rank, nranks));
    ctx->comm->a2a = alltoallv;
    ctx->comm->ag = allgather;
    ctx->comm->user = user;
    ctx->comm->copy = alltoallv == nullptr; // no all-to-all of the host's: the library's COPY transport over the host's all-gather
    return KMU_OK;
}

int kmu_comm_set_transport(kmu_ctx *ctx, int transport) {
    if (!ctx || !ctx->comm) return KMU_E_BAD_ARG;
    kmu_comm *c = ctx->comm;
    if (transport != KMU_TRANSPORT_DEFAULT && transport != KMU_TRANSPORT_COPY) return fail(ctx, KMU_E_BAD_ARG, "unknown transport %d", transport);
    if (c->copy_pending) KMU_TRY(comm_wait(ctx));
    if (transport == KMU_TRANSPORT_DEFAULT && !c->nccl && !c->a2a)
        return fail(ctx, KMU_E_BAD_ARG, "this communicator has no all-to-all but the COPY transport's");
    c->copy = transport == KMU_TRANSPORT_COPY;
    return KMU_OK;
}
int kmu_comm_transport(const kmu_ctx *ctx) { return ctx && ctx->comm && ctx->comm->copy ? KMU_TRANSPORT_COPY : KMU_TRANSPORT_DEFAULT; }

int kmu_comm_destroy(kmu_ctx *ctx) {
    if (!ctx) return KMU_E_BAD_ARG;
    comm_free(ctx);
    return KMU_OK;
}

int kmu_comm_rank(const kmu_ctx *ctx) { return ctx && ctx->comm ? ctx->comm->rank : -1; }
int kmu_comm_nranks(const kmu_ctx *ctx) { return ctx && ctx->comm ? ctx->comm->nranks : 0; }

int kmu_comm_get_stats(const kmu_ctx *ctx, kmu_comm_stats *out) {
    if (!ctx || !out || !ctx->comm) return KMU_E_BAD_ARG;
    kmu_comm *c = ctx->comm;
    (void) hipSetDevice(ctx->device);
    comm_collect_timed(c);
    c->stats.exchange_gbps_out = c->stats.exchange_ms > 0 ? (double) c->stats.bytes_sent / (c->stats.exchange_ms * 1e6) : 0.0;
    c->stats.exchange_gbps_in = c->stats.exchange_ms > 0 ? (double) c->stats.bytes_received / (c->stats.exchange_ms * 1e6) : 0.0;
    *out = c->stats;
    return KMU_OK;
}

// every rank hands in `bytes` of host memory and gets all ranks' contributions in rank order (what a host needs to agree
// on sizes before it hands buffers to the library; also the smallest complete test of a communicator)
int kmu_comm_allgather(kmu_ctx *ctx, const void *send_host, void *recv_host, uint64_t bytes) {
    if (!ctx || !send_host || !recv_host) return KMU_E_BAD_ARG;
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    return comm_allgather_host(ctx, send_host, recv_host, bytes);
}

} // extern "C"
