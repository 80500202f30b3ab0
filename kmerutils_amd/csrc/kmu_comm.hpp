// kmu_comm.hpp -- the communicator a kmu_ctx may carry: one rank per GPU (process or thread), RCCL over xGMI.
//
// RCCL is bound at run time (dlopen): a host that already carries an RCCL in its process -- PyTorch ships its own copy with
// the same SONAME -- must end up with ONE copy, the one already mapped; linking librccl.so.1 at build time would map a
// second one next to torch's (two sets of the same global symbols in one process).  A host-supplied transport (two function
// pointers) serves hosts that own a communicator already and the tests that run more ranks than the box has GPUs.
#pragma once

#include <utility>

#include "kmu_ctx.hpp"

struct kmu_comm {
    int rank = 0, nranks = 1;
    void *nccl = nullptr; // ncclComm_t
    kmu_alltoallv_fn a2a = nullptr;
    kmu_allgather_fn ag = nullptr;
    void *user = nullptr;
    hipStream_t stream = nullptr; // the exchange runs here, so that kernels on the context's stream can run under it
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    kmu_comm_stats stats{};
    // measured exchanges (kmu_comm_stats::exchange_ms): event pairs on the exchange stream not read yet, and the pool they return to
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;
    std::vector<hipEvent_t> ev_pool;
    // COPY transport (kmu_comm_set_transport(KMU_TRANSPORT_COPY), or a custom communicator without an all-to-all function): the
    // all-to-all is N - 1 device copies straight into the peers' receive buffers -- mapped into this process through HIP IPC, or
    // used as they are when the peer is a thread of this process -- on the exchange stream: no kernel of a collective library has
    // to find a CU under the persistent sketch kernels; the host's all-gather (or RCCL's) carries the handles and closes the exchange
    bool copy = false;
    bool copy_pending = false; // copies of an exchange are enqueued, the closing barrier has not run yet
    bool carrier_pending = false; // an exchange of the COPY transport went through the other data path (a rank could not export its buffer)
    struct Peer {
        hipIpcMemHandle_t handle{};
        void *base = nullptr; // the peer's receive buffer as this process sees it
        bool opened = false;  // through hipIpcOpenMemHandle (closed when the peer's buffer changes, and with the communicator)
    };
    std::vector<Peer> peers;
};

namespace kmu {

// every rank contributes `bytes` of host memory; recv gets nranks * bytes in rank order.  Synchronous.
int comm_allgather_host(kmu_ctx *ctx, const void *send, void *recv, uint64_t bytes);
// variable all-to-all of device memory in units of elem_bytes; counts / displacements per peer (elements).
// Enqueued on `s` (RCCL) or completed at return (host transport; `s` is synchronised first).
int comm_alltoallv(kmu_ctx *ctx, const void *send_dev, const uint64_t *send_counts, const uint64_t *send_displs, void *recv_dev,
                   const uint64_t *recv_counts, const uint64_t *recv_displs, uint32_t elem_bytes, hipStream_t s);

// what was received by the last comm_alltoallv is complete and visible to work enqueued on the context's stream from now on
// (RCCL: a stream dependency on the exchange stream; COPY transport: a host barrier over the ranks)
int comm_wait(kmu_ctx *ctx);

// zero the statistics for a new distributed add (event pairs not read yet are dropped)
void comm_stats_reset(kmu_ctx *ctx);

} // namespace kmu
