"""ctypes binding of libkmu.so (the C-ABI of include/kmu.h).

The product path: every compute call goes to hand-written gfx950 kernels.  There is no CPU fallback -- if the
shared library is missing or no HIP device is present this module raises, loudly.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
# KMU_LIB names another build of the library (A/B experiments: scripts/ab_libs.sh, scripts/pmc_lib.sh) -- the product file
# is never overwritten by an experiment
SO_PATH = os.environ.get("KMU_LIB") or os.path.join(_HERE, "libkmu.so")
_lib = None

# every symbol include/kmu.h declares (tests check the library exports all of them)
SYMBOLS = [
    "kmu_version", "kmu_device_count", "kmu_create", "kmu_destroy", "kmu_last_error", "kmu_synchronize", "kmu_stream",
    "kmu_profile_enable", "kmu_profile_reset", "kmu_profile_get", "kmu_count_non_acgt", "kmu_pack2b",
    "kmu_kmer_hashes", "kmu_sketch", "kmu_block_layout", "kmu_sketch_hashed", "kmu_count_create", "kmu_count_destroy",
    "kmu_count_reset", "kmu_count_add_reads", "kmu_count_add_kmers", "kmu_count_query", "kmu_count_nb_distinct",
    "kmu_count_nb_unique", "kmu_count_dump", "kmu_count_export_part", "kmu_count_merge_entries",
    "kmu_count_retain_part", "kmu_count_extract_by_owner", "kmu_sig_equal_pairs", "kmu_sig_equal_matrix",
    "kmu_minhash_distance_pairs", "kmu_ingest_fastq", "kmu_ingest_fasta", "kmu_ingest_fastx", "kmu_dev_alloc", "kmu_dev_free",
    "kmu_copy_to_device", "kmu_copy_to_host", "kmu_count_once_positions", "kmu_count_eliminate_once", "kmu_sketch_partial_words",
    "kmu_sketch_partial", "kmu_sketch_hashed_partial", "kmu_sketch_merge_partials", "kmu_kmer_hashes_compact", "kmu_set_hll_params",
    "kmu_kmer_hashes_range", "kmu_kmer_distribution", "kmu_nthash",
    "kmu_comm_get_id", "kmu_comm_init", "kmu_comm_init_custom", "kmu_comm_set_transport", "kmu_comm_transport", "kmu_comm_destroy", "kmu_comm_rank", "kmu_comm_nranks",
    "kmu_comm_allgather", "kmu_comm_get_stats", "kmu_count_finalize", "kmu_kmer_owner",
    "kmu_sketch_count", "kmu_host_alloc", "kmu_host_free", "kmu_count_nb_occurrences", "kmu_count_table_info",
    "kmu_count_nb_saturated", "kmu_kmer_owner_minimizer", "kmu_count_owner_kind", "kmu_count_extract_superkmers", "kmu_count_add_superkmers",
]


class KmuError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s: %s" % (A.STATUS_NAMES.get(code, str(code)), msg))
        self.code = code


def load():
    """Load libkmu.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError("kmerutils_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
    L = C.CDLL(SO_PATH)
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    L.kmu_version.restype = C.c_char_p
    L.kmu_device_count.restype = C.c_int
    L.kmu_create.argtypes = [C.POINTER(A.DeviceCfg), C.POINTER(vp)]
    L.kmu_destroy.argtypes = [vp]
    L.kmu_destroy.restype = None
    L.kmu_last_error.argtypes = [vp]
    L.kmu_last_error.restype = C.c_char_p
    L.kmu_synchronize.argtypes = [vp]
    L.kmu_stream.argtypes = [vp]
    L.kmu_stream.restype = vp
    L.kmu_profile_enable.argtypes = [vp, C.c_int]
    L.kmu_profile_reset.argtypes = [vp]
    L.kmu_profile_get.argtypes = [vp, C.POINTER(A.KernelStat), C.c_int]
    L.kmu_count_non_acgt.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, vp]
    L.kmu_pack2b.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, vp, vp]
    L.kmu_kmer_hashes.argtypes = [vp, C.POINTER(A.HashParams), vp, vp, vp, C.c_uint32, vp]
    L.kmu_kmer_hashes_range.argtypes = [vp, C.POINTER(A.HashParams), vp, vp, vp, C.c_uint32, vp, vp, vp]
    L.kmu_kmer_distribution.argtypes = [vp, C.POINTER(A.HashParams), vp, vp, vp, C.c_uint32, vp, vp, C.c_uint64, vp, u64p]
    L.kmu_nthash.argtypes = [vp, C.POINTER(A.NthashParams), vp, vp, vp, C.c_uint32, vp, vp]
    L.kmu_sketch_count.argtypes = [vp, C.POINTER(A.SketchParams), vp, vp, vp, C.c_uint32, vp]
    L.kmu_host_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.kmu_host_free.argtypes = [vp, vp]
    L.kmu_count_nb_occurrences.argtypes = [vp, u64p]
    L.kmu_count_nb_saturated.argtypes = [vp, u64p]
    L.kmu_count_table_info.argtypes = [vp, C.POINTER(A.CountTableInfo)]
    L.kmu_comm_get_id.argtypes = [C.POINTER(A.CommId)]
    L.kmu_comm_init.argtypes = [vp, C.POINTER(A.CommId), C.c_int, C.c_int]
    L.kmu_comm_init_custom.argtypes = [vp, C.c_int, C.c_int, A.ALLTOALLV_FN, A.ALLGATHER_FN, vp]
    L.kmu_comm_destroy.argtypes = [vp]
    L.kmu_comm_rank.argtypes = [vp]
    L.kmu_comm_nranks.argtypes = [vp]
    L.kmu_comm_allgather.argtypes = [vp, vp, vp, C.c_uint64]
    L.kmu_comm_get_stats.argtypes = [vp, C.POINTER(A.CommStats)]
    L.kmu_count_finalize.argtypes = [vp]
    L.kmu_kmer_owner.argtypes = [C.c_int, vp, C.c_uint64, C.c_uint32, vp]
    L.kmu_kmer_owner_minimizer.argtypes = [C.c_int, vp, C.c_uint64, C.c_uint32, vp]
    L.kmu_count_owner_kind.argtypes = [vp]
    L.kmu_count_extract_superkmers.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(vp), vp, vp]
    L.kmu_count_add_superkmers.argtypes = [vp, vp, C.c_uint64, C.c_int]
    L.kmu_sketch.argtypes = [vp, C.POINTER(A.SketchParams), vp, vp, vp, C.c_uint32, vp, vp, vp]
    L.kmu_block_layout.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.kmu_sketch_hashed.argtypes = [vp, C.POINTER(A.SketchParams), vp, vp, C.c_uint32, vp, vp]
    L.kmu_count_create.argtypes = [vp, C.POINTER(A.CountParams), C.POINTER(vp)]
    L.kmu_count_destroy.argtypes = [vp]
    L.kmu_count_destroy.restype = None
    L.kmu_count_reset.argtypes = [vp]
    L.kmu_count_add_reads.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_int, C.c_int]
    L.kmu_count_add_kmers.argtypes = [vp, vp, C.c_uint64, C.c_int]
    L.kmu_count_query.argtypes = [vp, vp, C.c_uint64, C.c_int, vp]
    L.kmu_count_nb_distinct.argtypes = [vp, u64p]
    L.kmu_count_nb_unique.argtypes = [vp, u64p]
    L.kmu_count_dump.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint64, u64p]
    L.kmu_count_export_part.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, C.c_uint64, C.c_int, u64p]
    L.kmu_count_merge_entries.argtypes = [vp, vp, vp, C.c_uint64, C.c_int]
    L.kmu_count_retain_part.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.kmu_count_extract_by_owner.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(vp), vp]
    L.kmu_sig_equal_pairs.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_int, vp, vp, C.c_uint64, C.c_int, vp]
    L.kmu_sig_equal_matrix.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp]
    L.kmu_minhash_distance_pairs.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_uint64, C.c_int, vp]
    L.kmu_set_hll_params.argtypes = [vp, C.POINTER(A.HllParams)]
    L.kmu_kmer_hashes_compact.argtypes = [vp, C.POINTER(A.HashParams), vp, vp, vp, C.c_uint32, vp, C.c_uint64, u64p]
    L.kmu_sketch_partial_words.argtypes = [C.POINTER(A.SketchParams)]
    L.kmu_sketch_partial_words.restype = C.c_uint32
    L.kmu_sketch_partial.argtypes = [vp, C.POINTER(A.SketchParams), vp, vp, vp, C.c_uint32, vp]
    L.kmu_sketch_hashed_partial.argtypes = [vp, C.POINTER(A.SketchParams), vp, vp, C.c_uint32, vp]
    L.kmu_sketch_merge_partials.argtypes = [vp, C.POINTER(A.SketchParams), vp, C.c_uint32, vp]
    L.kmu_count_eliminate_once.argtypes = [vp]
    L.kmu_count_once_positions.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, vp, vp, vp, C.c_uint64, u64p]
    L.kmu_dev_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.kmu_dev_free.argtypes = [vp, vp]
    L.kmu_copy_to_device.argtypes = [vp, vp, vp, C.c_uint64]
    L.kmu_copy_to_host.argtypes = [vp, vp, vp, C.c_uint64]
    for f in (L.kmu_ingest_fastq, L.kmu_ingest_fasta, L.kmu_ingest_fastx):
        f.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, C.c_uint64, vp, C.c_uint64, vp, C.POINTER(A.IngestInfo)]
    _lib = L
    return L


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    """(pointer, is_device) of a numpy array, a torch tensor, or None"""
    if x is None:
        return None, False
    if _is_torch(x):
        assert x.is_contiguous()
        return C.c_void_p(x.data_ptr()), x.is_cuda
    assert x.flags["C_CONTIGUOUS"]
    return x.ctypes.data_as(C.c_void_p), False


class _StreamOrdered:
    """libkmu's entry points as seen by one Context: every call is first ordered behind torch's current stream.

    The library enqueues on the context's stream.  When that is not torch's current stream (a context that owns its
    stream), nothing orders it behind work torch has queued on its own stream -- the producers of the input tensors, and
    the zero-fill of an output tensor allocated a line earlier, which could otherwise land on top of the kernel's results.
    Doing it here, at the one place every call passes through, covers the buffers of every present and future method."""

    _PLAIN = ("kmu_last_error", "kmu_destroy", "kmu_stream", "kmu_version", "kmu_device_count", "kmu_create",
              "kmu_block_layout", "kmu_sketch_partial_words", "kmu_count_destroy", "kmu_comm_get_id", "kmu_comm_rank",
              "kmu_comm_nranks", "kmu_comm_get_stats", "kmu_kmer_owner", "kmu_comm_destroy")

    def __init__(self, lib, ctx):
        self._lib = lib
        self._ctx = ctx

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if name in self._PLAIN:
            return f
        ctx = self._ctx

        def call(*args):
            ctx._order_after_torch()
            return f(*args)
        setattr(self, name, call)
        return call


class Context:
    """kmu_ctx: one HIP device + stream.  Use from one thread at a time."""

    def __init__(self, device_id=0, stream=None, async_device=False):
        lib = load()
        self.h = None
        self.device_id = device_id
        self.L = _StreamOrdered(lib, self)
        cfg = A.DeviceCfg(device_id, 1 if async_device else 0, stream, 0)
        h = C.c_void_p()
        rc = lib.kmu_create(C.byref(cfg), C.byref(h))
        if rc:
            raise KmuError(rc, lib.kmu_last_error(None).decode())
        self.h = h
        self._stream = lib.kmu_stream(h) or 0

    def close(self):
        if getattr(self, "h", None):
            self.L.kmu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise KmuError(rc, self.L.kmu_last_error(self.h).decode())

    def synchronize(self):
        self._check(self.L.kmu_synchronize(self.h))

    @property
    def stream(self):
        return self._stream

    def set_hll_params(self, b=1.001, a=20.0, q=65534):
        """kmu_set_hll_params: SetSketchParams of the following ALGO_HLL sketches on this context"""
        hp = A.HllParams(b, a, q, 0)
        self._check(self.L.kmu_set_hll_params(self.h, C.byref(hp)))

    # ---- communicator (one rank per GPU; kmu.h "multi-GPU") ----
    @staticmethod
    def comm_get_id():
        """kmu_comm_get_id: 128 bytes that rank 0 creates and hands to every other rank"""
        L = load()
        cid = A.CommId()
        rc = L.kmu_comm_get_id(C.byref(cid))
        if rc:
            raise KmuError(rc, L.kmu_last_error(None).decode())
        return bytes(C.string_at(C.byref(cid), A.COMM_ID_BYTES))

    def comm_init(self, comm_id, rank, nranks):
        """kmu_comm_init: RCCL communicator over the ranks' GPUs (collective)"""
        cid = A.CommId()
        C.memmove(C.byref(cid), comm_id, A.COMM_ID_BYTES)
        self._check(self.L.kmu_comm_init(self.h, C.byref(cid), rank, nranks))

    def comm_init_custom(self, rank, nranks, alltoallv, allgather):
        """kmu_comm_init_custom: `alltoallv(send_ptr, send_counts, send_displs, recv_ptr, recv_counts, recv_displs,
        elem_bytes)` (device pointers as ints, counts as lists) and `allgather(bytes) -> bytes of all ranks` supplied by
        the caller; exceptions inside them are reported as KMU_E_RCCL."""
        def a2a(_user, sp, sc, sd, rp, rc_, rd, eb, _stream):
            try:
                alltoallv(sp or 0, [sc[i] for i in range(nranks)], [sd[i] for i in range(nranks)], rp or 0,
                          [rc_[i] for i in range(nranks)], [rd[i] for i in range(nranks)], eb)
                return 0
            except Exception:  # noqa: BLE001 -- must not unwind through C
                import traceback
                traceback.print_exc()
                return 1

        def ag(_user, sp, rp, nbytes):
            try:
                out = allgather(C.string_at(sp, nbytes))
                assert len(out) == nbytes * nranks
                C.memmove(rp, out, len(out))
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        # alltoallv None: the library's COPY transport (device copies into the peers' receive buffers) over the caller's all-gather
        self._comm_cbs = (A.ALLTOALLV_FN(a2a) if alltoallv is not None else C.cast(None, A.ALLTOALLV_FN), A.ALLGATHER_FN(ag))  # keep the trampolines alive
        self._check(self.L.kmu_comm_init_custom(self.h, rank, nranks, self._comm_cbs[0], self._comm_cbs[1], None))

    def comm_set_transport(self, transport):
        """kmu_comm_set_transport: A.TRANSPORT_DEFAULT (RCCL / the host's all-to-all) or A.TRANSPORT_COPY (IPC-mapped device copies)"""
        self._check(self.L.kmu_comm_set_transport(self.h, int(transport)))

    @property
    def comm_transport(self):
        return self.L.kmu_comm_transport(self.h)

    def comm_destroy(self):
        self._check(self.L.kmu_comm_destroy(self.h))

    @property
    def comm_rank(self):
        return self.L.kmu_comm_rank(self.h)

    @property
    def comm_nranks(self):
        return self.L.kmu_comm_nranks(self.h)

    def comm_allgather(self, payload):
        """kmu_comm_allgather: bytes from every rank, in rank order"""
        n = self.comm_nranks
        src = C.create_string_buffer(bytes(payload), len(payload))
        dst = C.create_string_buffer(len(payload) * n)
        self._check(self.L.kmu_comm_allgather(self.h, src, dst, len(payload)))
        return dst.raw

    def comm_stats(self):
        st = A.CommStats()
        self._check(self.L.kmu_comm_get_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in A.CommStats._fields_}

    # ---- profiling ----
    def profile_enable(self, on=True):
        self._check(self.L.kmu_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._check(self.L.kmu_profile_reset(self.h))

    def profile_get(self):
        arr = (A.KernelStat * 32)()
        n = self.L.kmu_profile_get(self.h, arr, 32)
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(min(n, 32))}

    def _order_after_torch(self):
        """Wait for what torch has queued on its current stream of this device, unless that IS the context's stream
        (then stream order does it).  Only when torch is loaded and has touched the GPU: host-only callers never pay."""
        import sys
        torch = sys.modules.get("torch")
        if torch is None or self.h is None or not torch.cuda.is_initialized():
            return
        cur = torch.cuda.current_stream(self.device_id)
        if cur.cuda_stream != self._stream:
            cur.synchronize()

    def _wait_producers(self, *xs):
        """Kept for callers of round 1: the ordering now happens inside every library call (_StreamOrdered)."""
        self._order_after_torch()

    # ---- helpers ----
    @staticmethod
    def _mem(*xs):
        dev = [d for (_, d) in (_ptr(x) for x in xs if x is not None)]
        if any(dev) and not all(dev):
            raise ValueError("all buffers of one call must live on the same side (host or device)")
        return A.MEM_DEVICE if dev and dev[0] else A.MEM_HOST

    def count_non_acgt(self, bases, offsets):
        n = len(offsets) - 1
        mem = self._mem(bases, offsets)
        if mem == A.MEM_DEVICE:
            import torch
            out = torch.zeros(max(n, 1), dtype=torch.int64, device=bases.device)
        else:
            out = np.zeros(max(n, 1), np.uint64)
        self._check(self.L.kmu_count_non_acgt(self.h, _ptr(bases)[0], _ptr(offsets)[0], n, mem, _ptr(out)[0]))
        return out[:n]

    def pack2b(self, bases, offsets):
        n = len(offsets) - 1
        L = np.diff(offsets.astype(np.int64))
        poff = np.zeros(n + 1, np.uint64)
        out = np.zeros(int(((L + 3) // 4).sum()) + 16, np.uint8)
        self._check(self.L.kmu_pack2b(self.h, _ptr(bases)[0], _ptr(offsets)[0], n, A.MEM_HOST, _ptr(out)[0],
                                      _ptr(poff)[0]))
        return out[:int(poff[-1])], poff

    def kmer_hashes(self, bases, offsets, kmer_type, k, fhash, input_kind=A.INPUT_ASCII, packed_offsets=None,
                    out=None):
        n = len(offsets) - 1
        mem = self._mem(bases, offsets)
        self._wait_producers(bases)
        hp = A.HashParams(kmer_type, k, fhash, input_kind, mem, 0)
        if out is None:
            if mem == A.MEM_DEVICE:
                import torch
                out = torch.zeros(max(int(offsets[-1].item()), 1), dtype=torch.int64, device=bases.device)
            else:
                out = np.zeros(max(int(offsets[-1]), 1), np.uint64)
        self._check(self.L.kmu_kmer_hashes(self.h, C.byref(hp), _ptr(bases)[0], _ptr(offsets)[0],
                                           _ptr(packed_offsets)[0], n, _ptr(out)[0]))
        return out

    def kmer_hashes_range(self, bases, offsets, kmer_type, k, fhash, range_begin, range_end, input_kind=A.INPUT_ASCII,
                          packed_offsets=None, out=None):
        """kmu_kmer_hashes_range: generate_kmer_pattern_in_range -- fhash of the k-mers inside bases [begin_i, end_i) of
        every sequence, at out[offsets[i] + p]; a bad range raises KMU_E_BAD_ARG (the reference's set_range Err)."""
        n = len(offsets) - 1
        mem = self._mem(bases, offsets, range_begin, range_end)
        hp = A.HashParams(kmer_type, k, fhash, input_kind, mem, 0)
        if out is None:
            if mem == A.MEM_DEVICE:
                import torch
                out = torch.zeros(max(int(offsets[-1].item()), 1), dtype=torch.int64, device=bases.device)
            else:
                out = np.zeros(max(int(offsets[-1]), 1), np.uint64)
        self._check(self.L.kmu_kmer_hashes_range(self.h, C.byref(hp), _ptr(bases)[0], _ptr(offsets)[0], _ptr(packed_offsets)[0],
                                                 n, _ptr(range_begin)[0], _ptr(range_end)[0], _ptr(out)[0]))
        return out

    def kmer_distribution(self, bases, offsets, kmer_type, k, fhash=A.FHASH_IDENTITY_RAW, input_kind=A.INPUT_ASCII,
                          packed_offsets=None):
        """kmu_kmer_distribution: generate_kmer_distribution -- (values, multiplicities, dist_offsets): the distinct fhash
        values of sequence i and their counts are [dist_offsets[i], dist_offsets[i + 1]) of the first two arrays."""
        n = len(offsets) - 1
        mem = self._mem(bases, offsets)
        hp = A.HashParams(kmer_type, k, fhash, input_kind, mem, 0)
        cnt = C.c_uint64(0)
        args = (self.h, C.byref(hp), _ptr(bases)[0], _ptr(offsets)[0], _ptr(packed_offsets)[0], n)
        self._check(self.L.kmu_kmer_distribution(*args, None, None, 0, None, C.byref(cnt)))
        if mem == A.MEM_DEVICE:
            import torch
            kk = torch.zeros(max(cnt.value, 1), dtype=torch.int64, device=bases.device)
            cc = torch.zeros(max(cnt.value, 1), dtype=torch.int32, device=bases.device)
            do = torch.zeros(n + 1, dtype=torch.int64, device=bases.device)
        else:
            kk, cc, do = np.zeros(max(cnt.value, 1), np.uint64), np.zeros(max(cnt.value, 1), np.uint32), np.zeros(n + 1, np.uint64)
        self._check(self.L.kmu_kmer_distribution(*args, _ptr(kk)[0], _ptr(cc)[0], cnt.value, _ptr(do)[0], C.byref(cnt)))
        return kk[:cnt.value], cc[:cnt.value], do

    def nthash(self, bases, offsets, k, n_hashes=1, mode=A.NTHASH_CANONICAL, table=A.NTHASH_TABLE_2B,
               input_kind=A.INPUT_ASCII, packed_offsets=None, want_strand=True, out=None, strand_out=None):
        """kmu_nthash: (hashes [n_bases, n_hashes] u64, strand [n_bases] u8) -- rows of positions that start no k-mer
        stay untouched (zero in freshly allocated outputs)."""
        n = len(offsets) - 1
        mem = self._mem(bases, offsets)
        np_ = A.NthashParams(k, table, mode, n_hashes, input_kind, mem)
        if mem == A.MEM_DEVICE:
            import torch
            nb = max(int(offsets[-1].item()), 1)
            if out is None:
                out = torch.zeros((nb, n_hashes), dtype=torch.int64, device=bases.device)
            if want_strand and strand_out is None:
                strand_out = torch.zeros(nb, dtype=torch.uint8, device=bases.device)
        else:
            nb = max(int(offsets[-1]), 1)
            if out is None:
                out = np.zeros((nb, n_hashes), np.uint64)
            if want_strand and strand_out is None:
                strand_out = np.zeros(nb, np.uint8)
        self._check(self.L.kmu_nthash(self.h, C.byref(np_), _ptr(bases)[0], _ptr(offsets)[0], _ptr(packed_offsets)[0], n,
                                      _ptr(out)[0], _ptr(strand_out)[0]))
        return (out, strand_out) if want_strand else out

    def kmer_hashes_compact(self, bases, offsets, kmer_type, k, fhash, input_kind=A.INPUT_ASCII, packed_offsets=None):
        """kmu_kmer_hashes_compact: fhash of every k-mer, sequence after sequence, no gaps (uint64 / int64 tensor)"""
        n = len(offsets) - 1
        mem = self._mem(bases, offsets)
        self._wait_producers(bases)
        hp = A.HashParams(kmer_type, k, fhash, input_kind, mem, 0)
        cnt = C.c_uint64(0)
        self._check(self.L.kmu_kmer_hashes_compact(self.h, C.byref(hp), _ptr(bases)[0], _ptr(offsets)[0], _ptr(packed_offsets)[0],
                                                   n, None, 0, C.byref(cnt)))
        if mem == A.MEM_DEVICE:
            import torch
            out = torch.zeros(max(cnt.value, 1), dtype=torch.int64, device=bases.device)
        else:
            out = np.zeros(max(cnt.value, 1), np.uint64)
        self._check(self.L.kmu_kmer_hashes_compact(self.h, C.byref(hp), _ptr(bases)[0], _ptr(offsets)[0], _ptr(packed_offsets)[0],
                                                   n, _ptr(out)[0], cnt.value, C.byref(cnt)))
        return out[:cnt.value]

    def block_layout(self, offsets_host, block_size):
        n = len(offsets_host) - 1
        out = np.zeros(n + 1, np.uint64)
        rc = self.L.kmu_block_layout(_ptr(offsets_host)[0], n, block_size, _ptr(out)[0])
        self._check(rc)
        return out

    def sketch(self, bases, offsets, params, packed_offsets=None, block_row_offsets=None, n_rows=None, out=None,
               counts_out=None, want_counts=False):
        """kmu_sketch.  Host (numpy) or device (torch cuda) buffers.  Returns sig [rows, m] (and counts)."""
        n = len(offsets) - 1
        mem = self._mem(bases, offsets)
        self._wait_producers(bases)
        p = A.SketchParams.from_buffer_copy(params)
        p.mem = mem
        m = p.sketch_size
        if p.block_size > 0:
            if block_row_offsets is None:
                if mem != A.MEM_HOST:
                    raise ValueError("device block sketching needs block_row_offsets (+ n_rows)")
                block_row_offsets = self.block_layout(np.ascontiguousarray(offsets, np.uint64), p.block_size)
            rows = int(n_rows if n_rows is not None else block_row_offsets[-1])
        else:
            rows = 1 if p.mode == A.MODE_ALL_SEQS else n
        if out is None:
            if mem == A.MEM_DEVICE:
                import torch
                tdt = {A.SIG_U32: torch.int32, A.SIG_U64: torch.int64, A.SIG_F32: torch.float32,
                       A.SIG_F64: torch.float64, A.SIG_U16: torch.int16}[p.sig_type]
                out = torch.zeros((max(rows, 1), m), dtype=tdt, device=bases.device)
                if want_counts and counts_out is None:
                    counts_out = torch.zeros((max(rows, 1), m), dtype=torch.int32, device=bases.device)
            else:
                out = np.zeros((max(rows, 1), m), dtype=A.SIG_NP[p.sig_type])
                if want_counts and counts_out is None:
                    counts_out = np.zeros((max(rows, 1), m), np.uint32)
        self._check(self.L.kmu_sketch(self.h, C.byref(p), _ptr(bases)[0], _ptr(offsets)[0], _ptr(packed_offsets)[0],
                                      n, _ptr(block_row_offsets)[0], _ptr(out)[0], _ptr(counts_out)[0]))
        out = out[:rows]
        if want_counts:
            return out, counts_out[:rows]
        return out

    def sketch_count(self, bases, offsets, params, counter=None, out=None):
        """kmu_sketch_count: the reads once, both results -- signature rows (returned) and, if `counter` is given, the
        k-mer counts of the same reads.  Host buffers (numpy / CPU torch tensors, best pinned): chunked upload overlapped
        with the kernels, rows downloaded under the count build.  Device buffers: both on the resident reads."""
        n = len(offsets) - 1
        mem = self._mem(bases, offsets)
        p = A.SketchParams.from_buffer_copy(params)
        p.mem = mem
        p.mode = A.MODE_PER_SEQ
        if out is None:
            if mem == A.MEM_DEVICE:
                import torch
                tdt = {A.SIG_U32: torch.int32, A.SIG_U64: torch.int64, A.SIG_F32: torch.float32,
                       A.SIG_F64: torch.float64, A.SIG_U16: torch.int16}[p.sig_type]
                out = torch.zeros((max(n, 1), p.sketch_size), dtype=tdt, device=bases.device)
            else:
                out = np.zeros((max(n, 1), p.sketch_size), dtype=A.SIG_NP[p.sig_type])
        self._check(self.L.kmu_sketch_count(self.h, C.byref(p), counter.h if counter is not None else None, _ptr(bases)[0],
                                            _ptr(offsets)[0], n, _ptr(out)[0]))
        return out[:n]

    def sketch_hashed(self, hashed, offsets, params, want_counts=False):
        """kmu_sketch_hashed: the caller evaluated fhash; `hashed` holds Kmer::Val values (uint32 / uint64) of
        sequence i in hashed[offsets[i]:offsets[i+1]]."""
        n = len(offsets) - 1
        mem = self._mem(hashed, offsets)
        self._wait_producers(hashed)
        p = A.SketchParams.from_buffer_copy(params)
        p.mem = mem
        m = p.sketch_size
        rows = 1 if p.mode == A.MODE_ALL_SEQS else n
        counts = None
        if mem == A.MEM_DEVICE:
            import torch
            tdt = {A.SIG_U32: torch.int32, A.SIG_U64: torch.int64, A.SIG_F32: torch.float32,
                   A.SIG_F64: torch.float64, A.SIG_U16: torch.int16}[p.sig_type]
            out = torch.zeros((max(rows, 1), m), dtype=tdt, device=hashed.device)
            if want_counts:
                counts = torch.zeros((max(rows, 1), m), dtype=torch.int32, device=hashed.device)
        else:
            out = np.zeros((max(rows, 1), m), dtype=A.SIG_NP[p.sig_type])
            if want_counts:
                counts = np.zeros((max(rows, 1), m), np.uint32)
        self._check(self.L.kmu_sketch_hashed(self.h, C.byref(p), _ptr(hashed)[0], _ptr(offsets)[0], n, _ptr(out)[0],
                                             _ptr(counts)[0]))
        return (out[:rows], counts[:rows]) if want_counts else out[:rows]

    # ---- one signature over several GPUs: partial slot minima + merge ----
    def _partial_buf(self, p, like, n_parts=1):
        words = int(self.L.kmu_sketch_partial_words(C.byref(p)))
        if like is not None and _is_torch(like) and like.is_cuda:
            import torch
            return torch.zeros((n_parts, words), dtype=torch.int64, device=like.device)
        return np.zeros((n_parts, words), np.uint64)

    def sketch_partial(self, bases, offsets, params, packed_offsets=None):
        """kmu_sketch_partial: per-slot minima of these sequences (one row of kmu_sketch_partial_words words)"""
        mem = self._mem(bases, offsets)
        self._wait_producers(bases)
        p = A.SketchParams.from_buffer_copy(params)
        p.mem = mem
        out = self._partial_buf(p, bases)
        self._check(self.L.kmu_sketch_partial(self.h, C.byref(p), _ptr(bases)[0], _ptr(offsets)[0], _ptr(packed_offsets)[0],
                                              len(offsets) - 1, _ptr(out)[0]))
        return out[0]

    def sketch_hashed_partial(self, hashed, offsets, params):
        """kmu_sketch_hashed_partial: the same for caller-hashed values (disjoint key sets per rank for ProbMinHash)"""
        mem = self._mem(hashed, offsets)
        self._wait_producers(hashed)
        p = A.SketchParams.from_buffer_copy(params)
        p.mem = mem
        out = self._partial_buf(p, hashed)
        self._check(self.L.kmu_sketch_hashed_partial(self.h, C.byref(p), _ptr(hashed)[0], _ptr(offsets)[0], len(offsets) - 1,
                                                     _ptr(out)[0]))
        return out[0]

    def sketch_merge_partials(self, partials, params):
        """kmu_sketch_merge_partials: [n_parts, words] partial rows -> one signature row"""
        mem = self._mem(partials)
        self._wait_producers(partials)
        p = A.SketchParams.from_buffer_copy(params)
        p.mem = mem
        m = p.sketch_size
        if mem == A.MEM_DEVICE:
            import torch
            tdt = {A.SIG_U32: torch.int32, A.SIG_U64: torch.int64, A.SIG_F32: torch.float32, A.SIG_F64: torch.float64,
                   A.SIG_U16: torch.int16}[p.sig_type]
            out = torch.zeros(m, dtype=tdt, device=partials.device)
        else:
            out = np.zeros(m, dtype=A.SIG_NP[p.sig_type])
        self._check(self.L.kmu_sketch_merge_partials(self.h, C.byref(p), _ptr(partials)[0], int(partials.shape[0]), _ptr(out)[0]))
        return out

    # ---- ingest (kmu_ingest.hip) ----
    def ingest_fasta(self, text, want_index=False):
        """kmu_ingest_fasta: like ingest_fastq for a FASTA text (multi-line records)"""
        return self.ingest_fastq(text, want_index, fn="kmu_ingest_fasta")

    def ingest_fastx(self, text, want_index=False):
        """kmu_ingest_fastx: FASTA or FASTQ by the first byte (needletail::parse_fastx_file)"""
        return self.ingest_fastq(text, want_index, fn="kmu_ingest_fastx")

    def ingest_fastq(self, text, want_index=False, fn="kmu_ingest_fastq"):
        """kmu_ingest_fastq: FASTQ text (bytes / numpy uint8 on the host, torch uint8 tensor on the device) -> (bases,
        offsets, info[, record_index]) of the reads made of ACGTacgt only, in file order."""
        call = getattr(self.L, fn)
        if isinstance(text, (bytes, bytearray, memoryview)):
            text = np.frombuffer(bytes(text), np.uint8)
        dev = _is_torch(text) and text.is_cuda
        mem = A.MEM_DEVICE if dev else A.MEM_HOST
        self._wait_producers(text)
        n = int(text.shape[0])
        info = A.IngestInfo()
        self._check(call(self.h, _ptr(text)[0] if n else None, n, mem, None, 0, None, 0, None, C.byref(info)))
        if dev:
            import torch
            bases = torch.zeros(max(int(info.kept_bases), 1), dtype=torch.uint8, device=text.device)
            offs = torch.zeros(int(info.n_kept) + 1, dtype=torch.int64, device=text.device)
            idx = torch.zeros(max(int(info.n_kept), 1), dtype=torch.int32, device=text.device) if want_index else None
        else:
            bases = np.zeros(max(int(info.kept_bases), 1), np.uint8)
            offs = np.zeros(int(info.n_kept) + 1, np.uint64)
            idx = np.zeros(max(int(info.n_kept), 1), np.uint32) if want_index else None
        info2 = A.IngestInfo()
        self._check(call(self.h, _ptr(text)[0] if n else None, n, mem, _ptr(bases)[0], bases.shape[0], _ptr(offs)[0],
                         offs.shape[0], _ptr(idx)[0], C.byref(info2)))
        res = (bases[:int(info.kept_bases)], offs, info2)
        return res + (idx[:int(info.n_kept)],) if want_index else res

    # ---- signature comparison (kmu_compare.hip) ----
    def _new_like(self, ref, shape, np_dtype, torch_dtype_name):
        if _is_torch(ref) and ref.is_cuda:
            import torch
            return torch.zeros(shape, dtype=getattr(torch, torch_dtype_name), device=ref.device)
        return np.zeros(shape, np_dtype)

    @staticmethod
    def _sig_type_of(sig):
        name = str(sig.dtype).replace("torch.", "")
        return {"uint32": A.SIG_U32, "int32": A.SIG_U32, "uint64": A.SIG_U64, "int64": A.SIG_U64, "float32": A.SIG_F32,
                "float64": A.SIG_F64}[name]

    def sig_equal_pairs(self, sig_a, sig_b, ia, ib):
        """kmu_sig_equal_pairs: number of equal slots of rows sig_a[ia[p]] and sig_b[ib[p]]."""
        mem = self._mem(sig_a, sig_b, ia, ib)
        self._wait_producers(sig_a, sig_b)
        n = int(ia.shape[0])
        out = self._new_like(sig_a, max(n, 1), np.uint32, "int32")
        self._check(self.L.kmu_sig_equal_pairs(self.h, _ptr(sig_a)[0], sig_a.shape[0], _ptr(sig_b)[0], sig_b.shape[0],
                                               sig_a.shape[1], self._sig_type_of(sig_a), _ptr(ia)[0], _ptr(ib)[0], n, mem,
                                               _ptr(out)[0]))
        return out[:n]

    def sig_equal_matrix(self, sig_a, sig_b):
        """kmu_sig_equal_matrix: equal-slot counts of every row of sig_a against every row of sig_b."""
        mem = self._mem(sig_a, sig_b)
        self._wait_producers(sig_a, sig_b)
        out = self._new_like(sig_a, (max(sig_a.shape[0], 1), max(sig_b.shape[0], 1)), np.uint16, "int16")
        self._check(self.L.kmu_sig_equal_matrix(self.h, _ptr(sig_a)[0], sig_a.shape[0], _ptr(sig_b)[0], sig_b.shape[0],
                                                sig_a.shape[1], self._sig_type_of(sig_a), mem, _ptr(out)[0]))
        return out[:sig_a.shape[0], :sig_b.shape[0]]

    def minhash_distance_pairs(self, hashes_a, hashes_b, ia, ib):
        """kmu_minhash_distance_pairs on bottom-k rows: (common, total, i) per pair."""
        mem = self._mem(hashes_a, hashes_b, ia, ib)
        self._wait_producers(hashes_a, hashes_b)
        n = int(ia.shape[0])
        out = self._new_like(hashes_a, (max(n, 1), 3), np.uint32, "int32")
        self._check(self.L.kmu_minhash_distance_pairs(self.h, _ptr(hashes_a)[0], hashes_a.shape[0], _ptr(hashes_b)[0],
                                                      hashes_b.shape[0], hashes_a.shape[1], _ptr(ia)[0], _ptr(ib)[0], n,
                                                      mem, _ptr(out)[0]))
        return out[:n]

    def counter(self, kmer_type, k, counter_bits=8, capacity_hint=1 << 20, distributed=False, owner_hash=False, hint_occurrences=False):
        return Counter(self, kmer_type, k, counter_bits, capacity_hint, distributed, owner_hash, hint_occurrences)


class Counter:
    """kmu_counter: exact canonical k-mer multiplicities on the device (KmerCountT contract)."""

    def __init__(self, ctx, kmer_type, k, counter_bits=8, capacity_hint=1 << 20, distributed=False, owner_hash=False,
                 hint_occurrences=False):
        """hint_occurrences: capacity_hint counts k-mer occurrences; the first add sizes the table from the duplication it measures"""
        self.ctx = ctx
        self.L = ctx.L
        flags = ((A.COUNT_DISTRIBUTED if distributed else 0) | (A.COUNT_OWNER_HASH if owner_hash else 0)
                 | (A.COUNT_HINT_OCCURRENCES if hint_occurrences else 0))
        self.p = A.CountParams(kmer_type, k, counter_bits, flags, capacity_hint)
        h = C.c_void_p()
        ctx._check(self.L.kmu_count_create(ctx.h, C.byref(self.p), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.L.kmu_count_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.ctx._check(self.L.kmu_count_reset(self.h))

    def finalize(self):
        """kmu_count_finalize (collective): afterwards this rank's counter holds the k-mers it owns, counted over all ranks"""
        self.ctx._check(self.L.kmu_count_finalize(self.h))

    def add_reads(self, bases, offsets, input_kind=A.INPUT_ASCII, packed_offsets=None):
        mem = Context._mem(bases, offsets)
        self.ctx._wait_producers(bases)
        self.ctx._check(self.L.kmu_count_add_reads(self.h, _ptr(bases)[0], _ptr(offsets)[0], _ptr(packed_offsets)[0],
                                                   len(offsets) - 1, input_kind, mem))

    def add_kmers(self, canon):
        mem = Context._mem(canon)
        self.ctx._wait_producers(canon)
        n = canon.numel() if _is_torch(canon) else canon.size
        self.ctx._check(self.L.kmu_count_add_kmers(self.h, _ptr(canon)[0], n, mem))

    def query(self, canon):
        mem = Context._mem(canon)
        if mem == A.MEM_DEVICE:
            import torch
            n = canon.numel()
            out = torch.zeros(max(n, 1), dtype=torch.int32, device=canon.device)
        else:
            canon = np.ascontiguousarray(canon, np.uint64)
            n = canon.size
            out = np.zeros(max(n, 1), np.uint32)
        self.ctx._check(self.L.kmu_count_query(self.h, _ptr(canon)[0], n, mem, _ptr(out)[0]))
        return out[:n]

    def nb_distinct(self):
        v = C.c_uint64(0)
        self.ctx._check(self.L.kmu_count_nb_distinct(self.h, C.byref(v)))
        return v.value

    def nb_unique(self):
        v = C.c_uint64(0)
        self.ctx._check(self.L.kmu_count_nb_unique(self.h, C.byref(v)))
        return v.value

    def table_info(self):
        """kmu_count_table_info: {nslots, table_bytes, bytes_per_slot, count_field_bits}"""
        ti = A.CountTableInfo()
        self.ctx._check(self.L.kmu_count_table_info(self.h, C.byref(ti)))
        return {"nslots": ti.nslots, "table_bytes": ti.table_bytes, "bytes_per_slot": ti.bytes_per_slot,
                "count_field_bits": ti.count_field_bits, "count_ceiling": ti.count_ceiling}

    def table_bytes(self):
        return self.table_info()["table_bytes"]

    def nb_saturated(self):
        """kmu_count_nb_saturated: k-mers whose in-table count reached the table's ceiling (nb_occurrences is then a lower bound)"""
        n = C.c_uint64(0)
        self.ctx._check(self.L.kmu_count_nb_saturated(self.h, C.byref(n)))
        return n.value

    def nb_occurrences(self):
        """kmu_count_nb_occurrences: sum of the multiplicities held (= k-mer occurrences inserted)"""
        v = C.c_uint64(0)
        self.ctx._check(self.L.kmu_count_nb_occurrences(self.h, C.byref(v)))
        return v.value

    def dump(self, min_count=2):
        n = C.c_uint64(0)
        self.ctx._check(self.L.kmu_count_dump(self.h, min_count, None, None, 0, C.byref(n)))
        k = np.zeros(max(n.value, 1), np.uint64)
        c = np.zeros(max(n.value, 1), np.uint32)
        self.ctx._check(self.L.kmu_count_dump(self.h, min_count, _ptr(k)[0], _ptr(c)[0], n.value, C.byref(n)))
        return k[:n.value], c[:n.value]

    def eliminate_once(self):
        """kmu_count_eliminate_once: drop the k-mers seen once"""
        self.ctx._check(self.L.kmu_count_eliminate_once(self.h))

    def once_positions(self, bases, offsets):
        """kmu_count_once_positions: (canonical k-mers, numseq, numkmer) of the occurrences whose k-mer has count 1, in
        (sequence, position) order; numpy arrays for host input, torch cuda tensors for device input"""
        mem = Context._mem(bases, offsets)
        self.ctx._wait_producers(bases)
        n = C.c_uint64(0)
        nseq = len(offsets) - 1
        self.ctx._check(self.L.kmu_count_once_positions(self.h, _ptr(bases)[0], _ptr(offsets)[0], nseq, mem, None, None, None, 0,
                                                        C.byref(n)))
        if mem == A.MEM_DEVICE:
            import torch
            k = torch.zeros(max(n.value, 1), dtype=torch.int64, device=bases.device)
            s = torch.zeros(max(n.value, 1), dtype=torch.int32, device=bases.device)
            p = torch.zeros(max(n.value, 1), dtype=torch.int32, device=bases.device)
        else:
            k, s, p = np.zeros(max(n.value, 1), np.uint64), np.zeros(max(n.value, 1), np.uint32), np.zeros(max(n.value, 1), np.uint32)
        self.ctx._check(self.L.kmu_count_once_positions(self.h, _ptr(bases)[0], _ptr(offsets)[0], nseq, mem, _ptr(k)[0], _ptr(s)[0],
                                                        _ptr(p)[0], n.value, C.byref(n)))
        return k[:n.value], s[:n.value], p[:n.value]

    def export_part(self, part, n_parts, device=None):
        """entries owned by `part`: (kmers, counts) as numpy arrays, or torch cuda tensors when device is given"""
        n = C.c_uint64(0)
        self.ctx._check(self.L.kmu_count_export_part(self.h, part, n_parts, None, None, 0, A.MEM_HOST, C.byref(n)))
        if device is not None:
            import torch
            k = torch.zeros(max(n.value, 1), dtype=torch.int64, device=device)
            c = torch.zeros(max(n.value, 1), dtype=torch.int32, device=device)
            mem = A.MEM_DEVICE
        else:
            k = np.zeros(max(n.value, 1), np.uint64)
            c = np.zeros(max(n.value, 1), np.uint32)
            mem = A.MEM_HOST
        self.ctx._check(self.L.kmu_count_export_part(self.h, part, n_parts, _ptr(k)[0], _ptr(c)[0], n.value, mem,
                                                     C.byref(n)))
        return k[:n.value], c[:n.value]

    def extract_by_owner(self, bases, offsets, n_parts):
        """canonical k-mers of the reads grouped by owner rank.  Returns (kmers, bounds): `kmers` is a zero-copy view
        (torch int64 cuda tensor via __cuda_array_interface__) of a buffer owned by the context, valid until the next
        counter call; bounds[n_parts + 1] (numpy) delimits the groups."""
        import torch
        mem = Context._mem(bases, offsets)
        self.ctx._wait_producers(bases)
        ptr = C.c_void_p()
        bounds = np.zeros(n_parts + 1, np.uint64)
        self.ctx._check(self.L.kmu_count_extract_by_owner(self.h, _ptr(bases)[0], _ptr(offsets)[0], len(offsets) - 1, mem,
                                                          n_parts, C.byref(ptr), _ptr(bounds)[0]))
        n = int(bounds[-1])

        class _Dev:  # CUDA array interface v2 over the library-owned buffer
            __cuda_array_interface__ = {"shape": (max(n, 1),), "typestr": "<i8", "data": (ptr.value, False), "version": 2}

        dev = bases.device if _is_torch(bases) else torch.device("cuda", self.ctx.device_id)
        t = torch.as_tensor(_Dev(), device=dev)
        return t[:n], bounds

    @property
    def owner_kind(self):
        """kmu_count_owner_kind: A.OWNER_MINIMIZER / A.OWNER_HASH"""
        return int(self.L.kmu_count_owner_kind(self.h))

    def extract_superkmers(self, bases, offsets, n_parts):
        """the k-mers of the reads as super-k-mer records grouped by minimizer owner.  Returns (records, bounds, kmers):
        `records` a zero-copy [n, 3] int32 view (torch cuda tensor) of a buffer owned by the context, valid until the next
        counter call; bounds[n_parts + 1] (numpy, in records); kmers[n_parts] the k-mers every group holds."""
        import torch
        mem = Context._mem(bases, offsets)
        self.ctx._wait_producers(bases)
        ptr = C.c_void_p()
        bounds = np.zeros(n_parts + 1, np.uint64)
        kmers = np.zeros(n_parts, np.uint64)
        self.ctx._check(self.L.kmu_count_extract_superkmers(self.h, _ptr(bases)[0], _ptr(offsets)[0], len(offsets) - 1, mem, n_parts,
                                                            C.byref(ptr), _ptr(bounds)[0], _ptr(kmers)[0]))
        n = int(bounds[-1])

        class _Dev:  # CUDA array interface v2 over the library-owned buffer
            __cuda_array_interface__ = {"shape": (max(n, 1), 3), "typestr": "<i4", "data": (ptr.value, False), "version": 2}

        dev = bases.device if _is_torch(bases) else torch.device("cuda", self.ctx.device_id)
        t = torch.as_tensor(_Dev(), device=dev)
        return t[:n], bounds, kmers

    def add_superkmers(self, records):
        """kmu_count_add_superkmers: records as an [n, 3] int32 / uint32 array (numpy or torch cuda tensor)"""
        mem = Context._mem(records)
        self.ctx._wait_producers(records)
        n = (records.numel() if _is_torch(records) else records.size) // 3
        self.ctx._check(self.L.kmu_count_add_superkmers(self.h, _ptr(records)[0], n, mem))

    def merge_entries(self, kmers, counts):
        mem = Context._mem(kmers, counts)
        self.ctx._wait_producers(kmers)
        n = kmers.numel() if _is_torch(kmers) else kmers.size
        self.ctx._check(self.L.kmu_count_merge_entries(self.h, _ptr(kmers)[0], _ptr(counts)[0], n, mem))

    def retain_part(self, part, n_parts):
        self.ctx._check(self.L.kmu_count_retain_part(self.h, part, n_parts))


def kmer_owner(kmer_type, canon_kmers, n_parts):
    """kmu_kmer_owner: DispatchableT::dispatch of canonical k-mer values (host arithmetic)"""
    L = load()
    k = np.ascontiguousarray(canon_kmers, np.uint64)
    out = np.zeros(max(k.size, 1), np.uint32)
    rc = L.kmu_kmer_owner(kmer_type, _ptr(k)[0], k.size, n_parts, _ptr(out)[0])
    if rc:
        raise KmuError(rc, "kmu_kmer_owner")
    return out[:k.size]


def kmer_owner_minimizer(k, canon_kmers, n_parts):
    """kmu_kmer_owner_minimizer: the minimizer owner of Kmer64bit values of k bases (host arithmetic)"""
    L = load()
    v = np.ascontiguousarray(canon_kmers, np.uint64)
    out = np.zeros(max(v.size, 1), np.uint32)
    rc = L.kmu_kmer_owner_minimizer(k, _ptr(v)[0], v.size, n_parts, _ptr(out)[0])
    if rc:
        raise KmuError(rc, "kmu_kmer_owner_minimizer")
    return out[:v.size]
