"""ctypes mirror of include/kmu.h (enums + parameter structs).  Shared by the product loader (lib.py) and by
the test-only oracle loader (oracle/oracle.py); holds no compute."""
import ctypes as C

# kmu_status
OK = 0
E_BAD_K, E_BAD_ALPHABET, E_EMPTY_SEQ, E_NON_ACGT, E_OOM, E_HIP, E_BAD_ARG, E_TABLE_FULL, E_UNSUPPORTED, E_NO_DEVICE = (
    -1, -2, -3, -4, -5, -6, -7, -8, -9, -10)
STATUS_NAMES = {0: "KMU_OK", -1: "KMU_E_BAD_K", -2: "KMU_E_BAD_ALPHABET", -3: "KMU_E_EMPTY_SEQ", -4: "KMU_E_NON_ACGT",
                -5: "KMU_E_OOM", -6: "KMU_E_HIP", -7: "KMU_E_BAD_ARG", -8: "KMU_E_TABLE_FULL",
                -9: "KMU_E_UNSUPPORTED", -10: "KMU_E_NO_DEVICE", -11: "KMU_E_RCCL"}
E_RCCL = -11

MEM_HOST, MEM_DEVICE = 0, 1
INPUT_ASCII, INPUT_PACKED2 = 0, 1
KMER32BIT, KMER16B32BIT, KMER64BIT, KMERAA32BIT, KMERAA64BIT = 0, 1, 2, 3, 4
(FHASH_IDENTITY_RAW, FHASH_VALUE_MASKED, FHASH_CANON_RAW, FHASH_CANON_INVHASH, FHASH_INVHASH_RAW, FHASH_CANON_VALUE,
 FHASH_CANON_NTHASH, FHASH_CANON_NTHASH_8B) = range(8)
ALGO_PROB3A, ALGO_SUPER, ALGO_SUPER2, ALGO_BOTTOMK, ALGO_PROB3, ALGO_OPTDENS, ALGO_REVOPTDENS, ALGO_HLL = 0, 1, 2, 3, 4, 5, 6, 7
SIG_U32, SIG_U64, SIG_F32, SIG_F64, SIG_U16 = 0, 1, 2, 3, 4
HASHER_NOHASH, HASHER_FNV1A, HASHER_INT64HASH = 0, 1, 2
MODE_PER_SEQ, MODE_ALL_SEQS = 0, 1
FLAG_RAND08 = 0x1


def kmer_val_bytes(kmer_type):
    return 8 if kmer_type in (KMER64BIT, KMERAA64BIT) else 4


def kmer_type_for_k(k, aa=False):
    """The reference's choice of k-mer type for a k (kmer32bit.rs:68, kmer16b32bit.rs, kmer64bit.rs, kmeraa.rs)."""
    if aa:
        return KMERAA32BIT if k <= 6 else KMERAA64BIT
    if k <= 14:
        return KMER32BIT
    if k == 16:
        return KMER16B32BIT
    return KMER64BIT


class DeviceCfg(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("async_device", C.c_int32), ("stream", C.c_void_p),
                ("workspace_bytes", C.c_uint64)]


class SketchParams(C.Structure):
    _fields_ = [("algo", C.c_int32), ("kmer_type", C.c_int32), ("kmer_size", C.c_int32), ("sketch_size", C.c_int32),
                ("sig_type", C.c_int32), ("hasher", C.c_int32), ("fhash", C.c_int32), ("block_size", C.c_int32),
                ("mode", C.c_int32), ("input_kind", C.c_int32), ("mem", C.c_int32), ("flags", C.c_uint32)]


class HashParams(C.Structure):
    _fields_ = [("kmer_type", C.c_int32), ("kmer_size", C.c_int32), ("fhash", C.c_int32), ("input_kind", C.c_int32),
                ("mem", C.c_int32), ("flags", C.c_uint32)]


class CountParams(C.Structure):
    _fields_ = [("kmer_type", C.c_int32), ("kmer_size", C.c_int32), ("counter_bits", C.c_int32),
                ("flags", C.c_int32), ("capacity_hint", C.c_uint64)]


COUNT_DISTRIBUTED = 0x1
COUNT_OWNER_HASH = 0x2
COUNT_HINT_OCCURRENCES = 0x4
OWNER_HASH, OWNER_MINIMIZER = 0, 1
ROUTE_NONE, ROUTE_OCCURRENCES, ROUTE_MERGE, ROUTE_SUPERKMERS = 0, 1, 2, 3
SMER_REC_BYTES = 12
COMM_ID_BYTES = 128


class CommId(C.Structure):
    _fields_ = [("bytes", C.c_char * COMM_ID_BYTES)]


class CommStats(C.Structure):
    """kmu_comm_stats"""
    _fields_ = [("route", C.c_int32), ("sample_shift", C.c_int32), ("dup_ratio", C.c_double), ("kmers_local", C.c_uint64),
                ("bytes_occurrences", C.c_uint64), ("bytes_merge", C.c_uint64), ("bytes_sent", C.c_uint64),
                ("bytes_received", C.c_uint64), ("model_ms_occurrences", C.c_double), ("model_ms_merge", C.c_double),
                ("owner_kind", C.c_int32), ("exchanges", C.c_int32), ("records_local", C.c_uint64),
                ("exchange_ms", C.c_double), ("exchange_gbps_out", C.c_double), ("exchange_gbps_in", C.c_double)]


class CountTableInfo(C.Structure):
    """kmu_count_table_info_t"""
    _fields_ = [("nslots", C.c_uint64), ("table_bytes", C.c_uint64), ("bytes_per_slot", C.c_uint32),
                ("count_field_bits", C.c_uint32), ("count_ceiling", C.c_uint64)]


ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p,
                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)


class IngestInfo(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_kept", C.c_uint64), ("kept_bases", C.c_uint64), ("n_bases", C.c_uint64),
                ("nb_bad_bases", C.c_uint64), ("nb_bad_reads", C.c_uint64)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


SIG_NP = {SIG_U32: "uint32", SIG_U64: "uint64", SIG_F32: "float32", SIG_F64: "float64", SIG_U16: "uint16"}


class HllParams(C.Structure):
    """kmu_hll_params: SetSketchParams (b, a, q); m is the sketch size of the call"""
    _fields_ = [("b", C.c_double), ("a", C.c_double), ("q", C.c_uint32), ("reserved", C.c_uint32)]


NTHASH_CANONICAL, NTHASH_FORWARD, NTHASH_RCOMP = 0, 1, 2
NTHASH_TABLE_2B, NTHASH_TABLE_8B = 0, 1


class NthashParams(C.Structure):
    """kmu_nthash_params"""
    _fields_ = [("kmer_size", C.c_int32), ("table", C.c_int32), ("mode", C.c_int32), ("n_hashes", C.c_int32),
                ("input_kind", C.c_int32), ("mem", C.c_int32)]

# kmu_transport
TRANSPORT_DEFAULT, TRANSPORT_COPY = 0, 1
