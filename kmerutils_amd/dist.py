"""Multi-GPU orchestration: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).

Sketching shards reads -- independent units, exactly the reference's rayon map
(src/sketching/seqsketchjaccard.rs:245-248) -- with NO data-path collective: every rank sketches a contiguous
read range and the rows are concatenated in read order (the rank re-ordering of seqsketchjaccard.rs:250-259).

Counting has one real exchange step.  The reference partitions the *key space* over threads
(`int64_hash(kmer) % n`, src/base/kmercount.rs:412-420, :942) and ships every k-mer occurrence to its owner
through channels.  Across GPUs that exchange runs INSIDE libkmu.so (a distributed counter: kmu_count_add_reads /
kmu_count_finalize, include/kmu.h "multi-GPU"); this module only carries the communicator's 128-byte id over a torch
process group (`init_comm`) or lends the process group to the library as its transport (`TorchTransport`).  Either way rank r
ends with the exact counts of the keys with owner == r, i.e. the KmerCounterPool layout (kmercount.rs:424-565) with one
counter per GPU.  (A host that drives the exchange itself with the exported building blocks: tests/host_exchange.py.)

PyTorch is plumbing here: process group, all_to_all; the compute is in libkmu.
"""
import numpy as np


def shard_reads_by_bases(lens, world_size):
    """Contiguous read ranges [(r0, r1)] balanced by total bases (ONT lengths are skewed: balancing by read count
    would leave ranks idle).  Every read belongs to exactly one range; ranges may be empty."""
    lens = np.asarray(lens, dtype=np.int64)
    csum = np.concatenate([[0], np.cumsum(lens)])
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r // world_size
        bounds.append(int(np.searchsorted(csum, target, side="left")))
    bounds.append(len(lens))
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return [(bounds[i], bounds[i + 1]) for i in range(world_size)]


def _all_to_all_var(send_list, recv_sizes, dtype, device, group=None):
    """variable-size all-to-all of 1-D tensors; all_to_all_single on RCCL, isend/irecv pairs elsewhere (gloo has no
    all_to_all)"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    recv = [torch.empty(int(n), dtype=dtype, device=device) for n in recv_sizes]
    if dist.get_backend(group) == "nccl":
        sendbuf = torch.cat(send_list) if send_list else torch.empty(0, dtype=dtype, device=device)
        recvbuf = torch.empty(int(sum(recv_sizes)), dtype=dtype, device=device)
        dist.all_to_all_single(recvbuf, sendbuf, [int(n) for n in recv_sizes], [int(t.numel()) for t in send_list],
                               group=group)
        off = 0
        for i, n in enumerate(recv_sizes):
            recv[i] = recvbuf[off:off + int(n)]
            off += int(n)
        return recv
    reqs = []
    for peer in range(world):
        if peer == rank:
            recv[peer].copy_(send_list[peer])
            continue
        if send_list[peer].numel():
            reqs.append(dist.isend(send_list[peer].contiguous(), peer, group=group))
        if recv[peer].numel():
            reqs.append(dist.irecv(recv[peer], peer, group=group))
    for r in reqs:
        r.wait()
    return recv


def gather_rows(local_rows, group=None):
    """Concatenate per-rank signature slabs in rank (= read) order on every rank.  Used by the host API; the bench
    never needs it (each rank keeps its own slab)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=local_rows.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    mx = int(max(int(x.item()) for x in ns))
    pad = torch.zeros((mx,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    pad[:local_rows.shape[0]] = local_rows
    outs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:int(k.item())] for o, k in zip(outs, ns)], 0)


def sketch_seqs_distributed(ctx, bases, offsets, params, group=None):
    """`sketch_compressedkmer_seqs` over sequences held by several ranks: ONE signature for the union of every rank's reads,
    the same on every rank (and the same as one process would compute for the concatenated list).

    SuperMinHash / SuperMinHash2 / OptDens / RevOptDens treat k-mer occurrences independently: every rank reduces its reads
    to per-slot minima (`kmu_sketch_partial`), one all-gather of these m-word arrays, local merge.
    ProbMinHash3a weighs a key by its multiplicity over ALL reads, so per-rank minima of the local reads would be wrong:
    the fhash values of the local k-mers (`kmu_kmer_hashes_compact`) are exchanged by owner rank first -- one all-to-all, the
    exchange counting uses -- every rank sketches the disjoint key set it owns (`kmu_sketch_hashed_partial`), and the
    (h, key) minima are all-gathered and merged per slot.  `bases` / `offsets`: torch tensors on the context's device."""
    import torch
    import torch.distributed as dist
    from . import _abi as A
    p = A.SketchParams.from_buffer_copy(params)
    p.mode = A.MODE_ALL_SEQS
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return ctx.sketch(bases, offsets, p)[0]
    rank = dist.get_rank(group)
    nccl = dist.get_backend(group) == "nccl"
    dev = bases.device
    if p.algo in (A.ALGO_PROB3A, A.ALGO_PROB3):
        vals = ctx.kmer_hashes_compact(bases, offsets, p.kmer_type, p.kmer_size, p.fhash)
        ctx.synchronize()
        owner = torch.remainder((vals * -7046029254386353131) >> 33, world)  # any function of the value, the same on every rank
        order = torch.argsort(owner, stable=True)
        grouped = vals[order]
        counts = torch.bincount(owner, minlength=world).to(torch.int64)
        cdev = dev if nccl else torch.device("cpu")
        all_n = [torch.zeros(world, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(all_n, counts.to(cdev), group=group)
        recv_n = [int(all_n[q][rank].item()) for q in range(world)]
        bounds = torch.cumsum(counts, 0).cpu().tolist()
        send_list = [grouped[(bounds[q - 1] if q else 0):bounds[q]] for q in range(world)]
        if not nccl:  # gloo rehearsals stage the exchange through host memory
            send_list = [t.cpu() for t in send_list]
        recv = _all_to_all_var(send_list, recv_n, torch.int64, cdev, group)
        mine = (torch.cat(recv) if recv else torch.zeros(0, dtype=torch.int64)).to(dev)
        n_mine = int(mine.numel())
        if A.kmer_val_bytes(p.kmer_type) == 4:
            mine = mine.to(torch.int32)
        if n_mine == 0:
            mine = torch.zeros(1, dtype=mine.dtype, device=dev)
        off2 = torch.tensor([0, n_mine], dtype=torch.int64, device=dev)
        part = ctx.sketch_hashed_partial(mine.contiguous(), off2, p)
    else:
        part = ctx.sketch_partial(bases, offsets, p)
    ctx.synchronize()
    cpart = part if nccl else part.cpu()
    parts = [torch.zeros_like(cpart) for _ in range(world)]
    dist.all_gather(parts, cpart, group=group)
    allp = torch.stack(parts).to(dev)
    sig = ctx.sketch_merge_partials(allp.contiguous(), p)
    ctx.synchronize()
    return sig


# ---- the library's own communicator (kmu_comm_*, include/kmu.h "multi-GPU") ---------------------------------------------------
class TorchTransport:
    """The two functions kmu_comm_init_custom needs, carried by a torch process group: what a host that already owns a
    communicator hands to the library, and how the multi-rank paths are rehearsed on a box with fewer GPUs than ranks
    (gloo: staged through host memory).  `device` None: the pointers are host addresses (CPU tests of this class)."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.group = group
        self.device = device
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.nccl = dist.get_backend(group) == "nccl"

    def _view(self, ptr, nbytes):
        import torch
        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8, device=self.device or "cpu")
        if self.device is None:
            import ctypes as C
            return torch.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)

        class _Dev:
            __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

        return torch.as_tensor(_Dev(), device=self.device)

    def alltoallv(self, sp, sc, sd, rp, rc, rd, eb):
        import torch
        import torch.distributed as dist
        W = self.world
        send = self._view(sp, max((sd[p] + sc[p]) * eb for p in range(W)))
        recv = self._view(rp, max((rd[p] + rc[p]) * eb for p in range(W)))
        if self.nccl:
            contiguous = all(sd[p] == sum(sc[:p]) for p in range(W)) and all(rd[p] == sum(rc[:p]) for p in range(W))
            assert contiguous, "the library sends and receives contiguous groups in peer order"
            dist.all_to_all_single(recv[:sum(rc) * eb], send[:sum(sc) * eb], [n * eb for n in rc], [n * eb for n in sc],
                                   group=self.group)
            torch.cuda.synchronize(self.device)
            return
        reqs, staged = [], []
        for p in range(W):
            s = send[sd[p] * eb:(sd[p] + sc[p]) * eb]
            if p == self.rank:
                recv[rd[p] * eb:(rd[p] + rc[p]) * eb].copy_(s)
                continue
            if sc[p]:
                reqs.append(dist.isend(s.cpu().contiguous(), p, group=self.group))
            if rc[p]:
                t = torch.empty(rc[p] * eb, dtype=torch.uint8)
                staged.append((p, t))
                reqs.append(dist.irecv(t, p, group=self.group))
        for r in reqs:
            r.wait()
        for p, t in staged:
            recv[rd[p] * eb:(rd[p] + rc[p]) * eb].copy_(t)
        if self.device is not None:
            torch.cuda.synchronize(self.device)

    def allgather(self, payload):
        import torch
        import torch.distributed as dist
        dev = self.device if self.nccl else "cpu"
        mine = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(outs, mine, group=self.group)
        return b"".join(bytes(o.cpu().numpy().tobytes()) for o in outs)


class ThreadGroup:
    """Shared state of N ranks that are THREADS of one process (each with its own kmu context, all on one device): a host of
    that shape -- one process driving several contexts -- hands the library a transport of this kind through
    kmu_comm_init_custom.  It is also how the N-rank control flow (route agreement over N rows, N-way owner grouping, the
    finalize of MERGE among N owners) is exercised on a box with one GPU and a limit on the processes that may hold it."""

    def __init__(self, world):
        import threading
        self.world = world
        # (a rank that fails aborts the barrier -- ThreadTransport -- and nobody waits for ever: a broken or timed-out barrier
        # raises in every other rank's call, which the library reports as KMU_E_RCCL)
        self.barrier = threading.Barrier(world, timeout=600)
        self.a2a = [None] * world
        self.ag = [None] * world


class ThreadTransport:
    def __init__(self, group, rank, device):
        self.g, self.rank, self.device = group, rank, device

    def _view(self, ptr, nbytes):
        import torch
        if nbytes == 0 or not ptr:
            return torch.empty(0, dtype=torch.uint8, device=self.device)

        class _Dev:
            __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

        return torch.as_tensor(_Dev(), device=self.device)

    def alltoallv(self, sp, sc, sd, rp, rc, rd, eb):
        import torch
        g, W = self.g, self.g.world
        try:
            g.a2a[self.rank] = (sp, list(sc), list(sd))
            g.barrier.wait()  # every rank's send buffer is complete (the library synchronised its stream before the call)
            recv = self._view(rp, max((rd[p] + rc[p]) * eb for p in range(W)))
            for p in range(W):
                psp, psc, psd = g.a2a[p]
                if psc[self.rank] != rc[p]:
                    raise RuntimeError("rank %d expects %d items from %d, which sends %d" % (self.rank, rc[p], p, psc[self.rank]))
                if rc[p]:
                    src = self._view(psp, (psd[self.rank] + psc[self.rank]) * eb)
                    recv[rd[p] * eb:(rd[p] + rc[p]) * eb].copy_(src[psd[self.rank] * eb:(psd[self.rank] + psc[self.rank]) * eb])
            torch.cuda.synchronize(self.device)
            g.barrier.wait()  # nobody's send buffer is reused before every peer has pulled from it
        except BaseException:
            g.barrier.abort()  # the peers are (or will be) waiting at a barrier this rank never reaches: they fail instead of hanging
            raise

    def allgather(self, payload):
        g = self.g
        try:
            g.ag[self.rank] = bytes(payload)
            g.barrier.wait()
            out = b"".join(g.ag)
            g.barrier.wait()
        except BaseException:
            g.barrier.abort()
            raise
        return out


def init_comm_threads(ctx, group, rank, copy=False):
    """communicator of rank `rank` of a ThreadGroup on `ctx` (kmu_comm_init_custom); copy: the library's COPY transport carries
    the all-to-all (the ranks are threads of one process: a peer's receive buffer is used through its own address)"""
    import torch
    tt = ThreadTransport(group, rank, torch.device("cuda", ctx.device_id))
    ctx._transport = tt
    ctx.comm_init_custom(rank, group.world, None if copy else tt.alltoallv, tt.allgather)
    return tt


def init_comm(ctx, group=None, transport=None):
    """Give `ctx` (kmerutils_amd.lib.Context) a communicator over the ranks of a torch process group.

    transport "rccl" (the default whenever the group's backend is nccl, or there is no group): the library's own RCCL
    communicator -- rank 0 makes the id (kmu_comm_get_id), the 128 bytes travel over the process group, every rank calls
    kmu_comm_init; from then on the exchanges of distributed counters run inside libkmu.so, which is what a Rust host
    gets.  transport "copy": the same communicator with the library's COPY transport (kmu_comm_set_transport: the all-to-all as
    device copies into the peers' IPC-mapped receive buffers; one node).  transport "torch": the process group carries the
    exchanges (TorchTransport; gloo rehearsals); "copy+torch": the COPY transport with the process group's all-gather."""
    import torch
    import torch.distributed as dist
    from . import lib
    if not dist.is_initialized():
        ctx.comm_init(lib.Context.comm_get_id(), 0, 1)
        if transport == "copy":
            from . import _abi as A
            ctx.comm_set_transport(A.TRANSPORT_COPY)
            return "copy"
        return "rccl"
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if transport is None:
        transport = "rccl" if dist.get_backend(group) == "nccl" else "torch"
    if transport in ("rccl", "copy"):
        box = [lib.Context.comm_get_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ctx.comm_init(box[0], rank, world)
        if transport == "copy":  # the library's copy transport: RCCL carries the handles and the closing barrier, the copy engines the data
            from . import _abi as A
            ctx.comm_set_transport(A.TRANSPORT_COPY)
    elif transport == "copy+torch":  # the copy transport over the process group's all-gather (ranks that share a GPU: RCCL refuses those)
        tt = TorchTransport(group, torch.device("cuda", ctx.device_id))
        ctx._transport = tt
        # (the group's all-to-all stays the communicator's other data path: an exchange whose buffers cannot be exported takes it)
        ctx.comm_init_custom(rank, world, tt.alltoallv, tt.allgather)
        from . import _abi as A
        ctx.comm_set_transport(A.TRANSPORT_COPY)
    else:
        tt = TorchTransport(group, torch.device("cuda", ctx.device_id))
        ctx._transport = tt
        ctx.comm_init_custom(rank, world, tt.alltoallv, tt.allgather)
    return transport


def count_reads_distributed(counter, bases, offsets):
    """One collective step of distributed counting on a counter created with `distributed=True` on a context that has a
    communicator: this rank's shard in (kmu_count_add_reads picks OCCURRENCES or MERGE from the measured duplication),
    kmu_count_finalize, and the counter holds the k-mers this rank owns with their global multiplicities.  Returns the
    communicator's statistics of the step."""
    counter.add_reads(bases, offsets)
    counter.finalize()
    return counter.ctx.comm_stats()
