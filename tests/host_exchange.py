"""Test helper: a host that drives the exchange of distributed counting ITSELF with the library's building blocks
(kmu_count_extract_by_owner / kmu_count_add_kmers, kmu_count_export_part / kmu_count_merge_entries) over a torch process group --
round 1's Python exchange.  The product's N-rank path is the distributed counter inside libkmu.so (kmu_count_add_reads +
kmu_count_finalize, tests/test_gpu_comm.py); these two functions stay as tests of the exported building blocks."""
import numpy as np

from kmerutils_amd.dist import _all_to_all_var


def count_reads_exchange(counter, bases, offsets, group=None, overlap=None):
    """Distributed counting of this rank's read shard -- the throughput path.

    The reference's one-to-many driver dispatches every canonical k-mer to the thread that owns it,
    `int64_hash(kmer) % n` (src/base/kmercount.rs:933-949, :412-420).  Same dispatch here, across GPUs: the k-mers of
    the local reads are grouped by owner rank on the device (`kmu_count_extract_by_owner`), ONE all-to-all over
    RCCL/xGMI moves every group to its owner, and the owner builds its table from what it receives with the
    radix-partitioned build (`kmu_count_add_kmers`).  Afterwards rank r holds the exact global counts of the keys it
    owns: the KmerCounterPool layout with one counter per GPU.  Returns the number of k-mers received.

    `overlap`: optional callable run while the all-to-all is in flight (the exchange is xGMI traffic, the sketch
    kernel is ALU work: bench.py sketches the same reads under it).  It must not touch the counter."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        if overlap is not None:
            overlap()
        counter.add_reads(bases, offsets)
        return 0
    rank = dist.get_rank(group)
    kmers, bounds = counter.extract_by_owner(bases, offsets, world)
    kmers = torch.as_tensor(kmers)
    dev = kmers.device
    nccl = dist.get_backend(group) == "nccl"
    cdev = dev if nccl else torch.device("cpu")
    send_n = torch.as_tensor(np.diff(bounds.astype(np.int64)), dtype=torch.int64).to(cdev)
    all_n = [torch.zeros(world, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(all_n, send_n, group=group)
    recv_n = [int(all_n[p][rank].item()) for p in range(world)]
    send_sizes = [int(bounds[p + 1]) - int(bounds[p]) for p in range(world)]
    if nccl:
        # the groups are already contiguous in owner order: send straight from the library's buffer
        flat = torch.empty(int(sum(recv_n)), dtype=torch.int64, device=dev)
        work = dist.all_to_all_single(flat, kmers, recv_n, send_sizes, group=group, async_op=True)
        if overlap is not None:
            overlap()
        work.wait()
    else:
        # gloo (CPU tests, single-GPU rehearsals): point-to-point pairs, staged through host memory
        send_list = [kmers[int(bounds[p]):int(bounds[p + 1])].cpu() for p in range(world)]
        recv = [torch.empty(int(n), dtype=torch.int64) for n in recv_n]
        reqs = []
        for peer in range(world):
            if peer == rank:
                recv[peer].copy_(send_list[peer])
                continue
            if send_list[peer].numel():
                reqs.append(dist.isend(send_list[peer].contiguous(), peer, group=group))
            if recv[peer].numel():
                reqs.append(dist.irecv(recv[peer], peer, group=group))
        if overlap is not None:
            overlap()
        for r in reqs:
            r.wait()
        flat = (torch.cat(recv) if recv else torch.empty(0, dtype=torch.int64)).to(dev)
    if flat.numel():
        counter.add_kmers(flat if dev.type != "cpu" else flat.numpy().view(np.uint64))
    return int(flat.numel())


def merge_counters(counter, device=None, group=None, chunk_entries=1 << 27):
    """The exchange step of distributed counting.  `counter` is a kmerutils_amd.lib.Counter (or any object with
    export_part / reset / merge_entries).  On return the local table holds exactly the keys owned by this rank with
    their global multiplicities.  Returns the number of entries received."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return 0
    parts = [counter.export_part(p, world, device) for p in range(world)]
    tk = [torch.as_tensor(k.view(np.int64) if isinstance(k, np.ndarray) else k) for k, _ in parts]
    tc = [torch.as_tensor(c.view(np.int32) if isinstance(c, np.ndarray) else c) for _, c in parts]
    dev = tk[0].device
    send_n = torch.tensor([t.numel() for t in tk], dtype=torch.int64, device=dev)
    all_n = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_n, send_n, group=group)
    recv_n = [int(all_n[p][rank].item()) for p in range(world)]
    max_pair = int(max(int(x.max().item()) for x in all_n))
    counter.reset()
    received = 0
    # chunked so that no single collective moves more than chunk_entries per peer
    for c0 in range(0, max(max_pair, 1), chunk_entries):
        sk = [t[c0:c0 + chunk_entries] for t in tk]
        sc = [t[c0:c0 + chunk_entries] for t in tc]
        rn = [max(0, min(n - c0, chunk_entries)) for n in recv_n]
        rk = _all_to_all_var(sk, rn, torch.int64, dev, group)
        rc = _all_to_all_var(sc, rn, torch.int32, dev, group)
        for k, c in zip(rk, rc):
            if k.numel():
                if dev.type == "cpu":
                    counter.merge_entries(k.numpy().view(np.uint64), c.numpy().view(np.uint32))
                else:
                    counter.merge_entries(k.contiguous(), c.contiguous())
                received += int(k.numel())
    return received
