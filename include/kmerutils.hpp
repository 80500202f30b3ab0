/*
 * kmerutils.hpp -- host side above the C-ABI (include/kmu.h), in C++ because the reference is compiled code (Rust) whose
 * toolchain is absent from this image.  It mirrors the reference's own interface for the hot path -- same type and method
 * names, same argument meaning, errors where the reference panics -- so that code written against jean-pierreBoth/kmerutils
 * (and the tests in tests/cpp, which follow the reference's tests) reads the same.  Header-only, C++17, links libkmu.so only.
 *
 *   base          Sequence                                   src/base/sequence.rs:14-320
 *                 Kmer32bit / Kmer16b32bit / Kmer64bit       src/base/kmer32bit.rs, kmer16b32bit.rs, kmer64bit.rs
 *                 KmerT / CompressedKmerT / KmerBuilder      src/base/kmertraits.rs:14-52
 *                 KmerGenerator::generate_kmer               src/base/kmergenerator.rs:117-239
 *   aautils       SequenceAA, KmerAA32bit, KmerAA64bit       src/aautils/kmeraa.rs:146-484
 *   sketching     SeqSketcherParams, SketchAlgo, DataType    src/sketcharg.rs:13-78
 *                 SeqSketcherT (trait)                       src/sketching/setsketchert.rs:54-80
 *                 ProbHash3aSketch / SuperHashSketch / SuperHash2Sketch   setsketchert.rs:85-336, 904-1046
 *                 OptDensHashSketch / RevOptDensHashSketch                setsketchert.rs:343-599
 *                 HyperLogLogSketch, SetSketchParams, HllSeqsThreading    setsketchert.rs:600-896
 *                 SeqSketcher                                src/sketching/seqsketchjaccard.rs:117-415
 *                 jaccard_index_probminhash3a, probminhash_get_jaccard_objects, compute_*_jaccard
 *                                                            seqsketchjaccard.rs:58-108, 423-495
 *                 BlockSeqSketcher, BlockSketched, BlockSketchedSeq, DistBlockSketched
 *                                                            src/sketching/seqblocksketch.rs:38-227, 419-440
 *                 MinInvHashCountKmer, minhash_distance      src/sketching/minhash.rs:134-340
 *   counting      KmerCountT (trait), KmerCounter, KmerCounterPool, count_kmer_threaded_one_to_many
 *                                                            src/base/kmercount.rs:48-123, 424-565, 881-974
 *   io            FASTQ reader rule, signature / count dumps src/io.rs:12-72, src/bin/datasketcher.rs:358-388,
 *                                                            seqsketchjaccard.rs:385-414, kmercount.rs:467-531
 *
 * What runs where: containers and bookkeeping on the host (as upstream); k-mer generation, canonicalisation, hashing,
 * multisets, sketches, counting, FASTQ filtering and signature comparison on the GPU through libkmu.  There is no CPU
 * implementation of those steps here: without libkmu.so / a device every call throws KmuError.
 *
 * A Rust closure `fhash: Fn(&Kmer) -> Kmer::Val` cannot cross an FFI.  Two ways to say what it is:
 *   - an FHash constant naming one of the closures the reference's own callers use (evaluated on the device), e.g.
 *     `kmer_revcomp_hash_fn` for `|kmer| intNN_hash(kmer.reverse_complement().min(*kmer).0)`;
 *   - any C++ callable `Val(const Kmer&)`: the k-mers are generated on the device, the callable is evaluated on the host,
 *     multiset + sketch run on the device (kmu_sketch_hashed, the fallback of include/kmu.h).
 */
#ifndef KMERUTILS_HPP
#define KMERUTILS_HPP

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "kmu.h"

namespace kmerutils {

// =====================================================================================================================
// errors + device context
// =====================================================================================================================

/// What the reference reports by panicking (bad k, empty sequence, non-ACGT byte ...) arrives here as an exception.
class KmuError : public std::runtime_error {
  public:
    KmuError(int status, const std::string &what) : std::runtime_error(what), status_(status) {}
    int status() const { return status_; }

  private:
    int status_;
};

/// One HIP device + stream (kmu_ctx).  Use from one thread at a time.
class Context {
  public:
    explicit Context(int device_id = 0) {
        kmu_device_cfg cfg{};
        cfg.device_id = device_id;
        int rc = kmu_create(&cfg, &ctx_);
        if (rc != KMU_OK) throw KmuError(rc, std::string("kmu_create: ") + kmu_last_error(nullptr));
    }
    ~Context() {
        if (ctx_) kmu_destroy(ctx_);
    }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;

    kmu_ctx *raw() const { return ctx_; }
    void check(int rc) const {
        if (rc != KMU_OK) throw KmuError(rc, kmu_last_error(ctx_));
    }
    /// process-wide context on device 0, created on first use (the reference has no such object: rayon's global pool
    /// plays that role there)
    static Context &global() {
        static Context ctx(0);
        return ctx;
    }

  private:
    kmu_ctx *ctx_ = nullptr;
};

/// A buffer in the memory of the context's GPU (kmu_dev_alloc): for data that should cross PCIe once.
class DeviceBuffer {
  public:
    DeviceBuffer() = default;
    DeviceBuffer(Context &ctx, uint64_t bytes) : ctx_(&ctx), bytes_(bytes) { ctx.check(kmu_dev_alloc(ctx.raw(), bytes, &p_)); }
    ~DeviceBuffer() { release(); }
    DeviceBuffer(DeviceBuffer &&o) noexcept : ctx_(o.ctx_), p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept {
        if (this != &o) {
            release();
            ctx_ = o.ctx_; p_ = o.p_; bytes_ = o.bytes_;
            o.p_ = nullptr;
        }
        return *this;
    }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    void *data() const { return p_; }
    template <class T> T *as() const { return static_cast<T *>(p_); }
    uint64_t bytes() const { return bytes_; }
    void upload(const void *src, uint64_t n, uint64_t at = 0) { ctx_->check(kmu_copy_to_device(ctx_->raw(), as<uint8_t>() + at, src, n)); }
    void download(void *dst, uint64_t n, uint64_t from = 0) const { ctx_->check(kmu_copy_to_host(ctx_->raw(), dst, as<uint8_t>() + from, n)); }

  private:
    void release() {
        if (p_) (void) kmu_dev_free(ctx_->raw(), p_);
        p_ = nullptr;
    }
    Context *ctx_ = nullptr;
    void *p_ = nullptr;
    uint64_t bytes_ = 0;
};

// =====================================================================================================================
// base: Sequence
// =====================================================================================================================

/// 2-bit alphabet, src/base/alphabet.rs:119-127 (A0 C1 G2 T3, case-insensitive)
struct Alphabet2b {
    static bool is_valid_base(uint8_t c) {
        switch (c) {
        case 'A': case 'C': case 'G': case 'T': case 'a': case 'c': case 'g': case 't': return true;
        default: return false;
        }
    }
    static uint8_t encode(uint8_t c) {
        switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: throw std::invalid_argument("Alphabet2b::encode: not a base in ACGT");  // upstream panics
        }
    }
    static uint8_t decode(uint8_t code) { return "ACGT"[code & 3]; }
};

/// Sequence::new(raw, 2): 4 bases per byte, first base in bits 7..6, tail padded with A (sequence.rs:25-106).
/// Only the 2-bit representation is on the path (KmerSeqIterator refuses the others, kmergenerator.rs:221-223).
class Sequence {
  public:
    Sequence(const uint8_t *raw, size_t n, uint8_t nb_bits = 2) : nb_bases_(n) {
        if (nb_bits != 2) throw std::invalid_argument("Sequence: only 2 bits per base are supported on this path");
        seq_.assign((n + 3) / 4, 0);
        for (size_t i = 0; i < n; i++) seq_[i >> 2] |= uint8_t(Alphabet2b::encode(raw[i]) << (6 - 2 * (i & 3)));
    }
    explicit Sequence(std::string_view s, uint8_t nb_bits = 2)
        : Sequence(reinterpret_cast<const uint8_t *>(s.data()), s.size(), nb_bits) {}

    uint8_t nb_bits_by_base() const { return 2; }
    size_t size() const { return nb_bases_; }
    size_t compressed_length() const { return seq_.size(); }
    /// the 2-bit code of base `pos` (sequence.rs:120-139)
    uint8_t get_base(size_t pos) const { return (seq_[pos >> 2] >> (6 - 2 * (pos & 3))) & 3; }
    std::vector<uint8_t> decompress() const {
        std::vector<uint8_t> out(nb_bases_);
        for (size_t i = 0; i < nb_bases_; i++) out[i] = Alphabet2b::decode(get_base(i));
        return out;
    }
    Sequence get_reverse_complement() const {
        std::vector<uint8_t> rc(nb_bases_);
        for (size_t i = 0; i < nb_bases_; i++) rc[i] = Alphabet2b::decode(3 - get_base(nb_bases_ - 1 - i));
        return Sequence(rc.data(), rc.size(), 2);
    }
    const std::vector<uint8_t> &packed() const { return seq_; }

  private:
    std::vector<uint8_t> seq_;
    size_t nb_bases_;
};

// =====================================================================================================================
// base: k-mer value types (host-side value semantics; the device computes the same values, kmu_kmer_hashes)
// =====================================================================================================================

namespace detail {
inline uint32_t bitrev32(uint32_t x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}
inline uint64_t bitrev64(uint64_t x) { return (uint64_t(bitrev32(uint32_t(x))) << 32) | bitrev32(uint32_t(x >> 32)); }
/// complement every base and reverse their order inside the word: !v, reverse the bits, swap the bits of each pair
inline uint32_t revcomp_word(uint32_t v) {
    uint32_t r = bitrev32(~v);
    return ((r & 0x55555555u) << 1) | ((r & 0xAAAAAAAAu) >> 1);
}
inline uint64_t revcomp_word(uint64_t v) {
    uint64_t r = bitrev64(~v);
    return ((r & 0x5555555555555555ull) << 1) | ((r & 0xAAAAAAAAAAAAAAAAull) >> 1);
}
}  // namespace detail

/// up to 14 bases in a u32, k in bits 31..28 (kmer32bit.rs:22-217).  `v` is the reference's `.0`.
struct Kmer32bit {
    using Val = uint32_t;
    static constexpr int kmu_type = KMU_KMER32BIT;
    static constexpr bool is_aa = false;
    uint32_t v = 0;

    Kmer32bit() = default;
    explicit Kmer32bit(uint32_t raw) : v(raw) {}
    static Kmer32bit build(Val val, uint8_t k) { return Kmer32bit((uint32_t(k) << 28) | val); }   // KmerBuilder
    static Kmer32bit from_raw(uint64_t raw, uint8_t) { return Kmer32bit(uint32_t(raw)); }
    static constexpr size_t get_nb_base_max() { return 14; }
    static constexpr size_t get_bitsize() { return 32; }
    uint8_t get_nb_base() const { return uint8_t(v >> 28); }
    Val get_compressed_value() const { return v & 0x0FFFFFFFu; }
    Val raw() const { return v; }
    Kmer32bit push(uint8_t base) const {
        const uint32_t k = v >> 28;
        return Kmer32bit((v & 0xF0000000u) | (((v << 2) & ((1u << (2 * k)) - 1u)) | (base & 3u)));
    }
    Kmer32bit reverse_complement() const {
        const uint32_t k = v >> 28;
        return Kmer32bit((v & 0xF0000000u) | ((detail::revcomp_word(v) >> (32 - 2 * k)) & 0x0FFFFFFFu));
    }
    std::vector<uint8_t> get_uncompressed_kmer() const {
        const int k = get_nb_base();
        std::vector<uint8_t> s(k);
        for (int i = 0; i < k; i++) s[i] = Alphabet2b::decode(uint8_t(v >> (2 * (k - 1 - i))));
        return s;
    }
    /// Ord: number of bases first, then value (kmer32bit.rs:47-55)
    bool operator<(const Kmer32bit &o) const {
        return (v >> 28) != (o.v >> 28) ? (v >> 28) < (o.v >> 28) : get_compressed_value() < o.get_compressed_value();
    }
    bool operator==(const Kmer32bit &o) const { return v == o.v; }
    Kmer32bit min(const Kmer32bit &o) const { return o < *this ? o : *this; }
};

/// exactly 16 bases in a u32 (kmer16b32bit.rs:21-132)
struct Kmer16b32bit {
    using Val = uint32_t;
    static constexpr int kmu_type = KMU_KMER16B32BIT;
    static constexpr bool is_aa = false;
    uint32_t v = 0;

    Kmer16b32bit() = default;
    explicit Kmer16b32bit(uint32_t raw) : v(raw) {}
    static Kmer16b32bit build(Val val, uint8_t) { return Kmer16b32bit(val); }
    static Kmer16b32bit from_raw(uint64_t raw, uint8_t) { return Kmer16b32bit(uint32_t(raw)); }
    static constexpr size_t get_nb_base_max() { return 16; }
    static constexpr size_t get_bitsize() { return 32; }
    uint8_t get_nb_base() const { return 16; }
    Val get_compressed_value() const { return v; }
    Val raw() const { return v; }
    Kmer16b32bit push(uint8_t base) const { return Kmer16b32bit((v << 2) | (base & 3u)); }
    Kmer16b32bit reverse_complement() const { return Kmer16b32bit(detail::revcomp_word(v)); }
    std::vector<uint8_t> get_uncompressed_kmer() const {
        std::vector<uint8_t> s(16);
        for (int i = 0; i < 16; i++) s[i] = Alphabet2b::decode(uint8_t(v >> (2 * (15 - i))));
        return s;
    }
    bool operator<(const Kmer16b32bit &o) const { return v < o.v; }
    bool operator==(const Kmer16b32bit &o) const { return v == o.v; }
    Kmer16b32bit min(const Kmer16b32bit &o) const { return o < *this ? o : *this; }
};

/// up to 31 bases: (u64 value, u8 k) (kmer64bit.rs:24-172; k = 32 is broken upstream)
struct Kmer64bit {
    using Val = uint64_t;
    static constexpr int kmu_type = KMU_KMER64BIT;
    static constexpr bool is_aa = false;
    uint64_t v = 0;   // `.0`
    uint8_t k = 0;    // `.1`

    Kmer64bit() = default;
    Kmer64bit(uint64_t val, uint8_t nb) : v(val), k(nb) {}
    static Kmer64bit build(Val val, uint8_t nb) { return Kmer64bit(val, nb); }
    static Kmer64bit from_raw(uint64_t raw, uint8_t nb) { return Kmer64bit(raw, nb); }
    static constexpr size_t get_nb_base_max() { return 32; }
    static constexpr size_t get_bitsize() { return 64; }
    uint8_t get_nb_base() const { return k; }
    Val get_compressed_value() const { return v; }
    Val raw() const { return v; }
    Kmer64bit push(uint8_t base) const { return Kmer64bit(((v << 2) & ((1ull << (2 * k)) - 1ull)) | (base & 3ull), k); }
    Kmer64bit reverse_complement() const { return Kmer64bit(detail::revcomp_word(v) >> (64 - 2 * k), k); }
    std::vector<uint8_t> get_uncompressed_kmer() const {
        std::vector<uint8_t> s(k);
        for (int i = 0; i < k; i++) s[i] = Alphabet2b::decode(uint8_t(v >> (2 * (k - 1 - i))));
        return s;
    }
    bool operator<(const Kmer64bit &o) const { return k != o.k ? k < o.k : v < o.v; }
    bool operator==(const Kmer64bit &o) const { return v == o.v && k == o.k; }
    Kmer64bit min(const Kmer64bit &o) const { return o < *this ? o : *this; }
};

// ---- amino acids (src/aautils/kmeraa.rs) -----------------------------------------------------------------------------

/// aautils::kmeraa::Alphabet: 5 bits, upper case only (kmeraa.rs:29-139)
struct AlphabetAA {
    static constexpr uint8_t get_nb_bits() { return 5; }
    static bool is_valid_base(uint8_t c) { return code_of(c) != 0; }
    static uint8_t encode(uint8_t c) {
        const uint8_t code = code_of(c);
        if (!code) throw std::invalid_argument("aautils Alphabet::encode: not an amino acid letter");
        return code;
    }

  private:
    static uint8_t code_of(uint8_t c) {
        // A1 C2 D3 E4 F5 G6 H7 I8 K9 L10 M11 N12 P13 Q15 R16 S17 T18 V19 W20 Y21 (kmeraa.rs:86-107)
        static const char *letters = "ACDEFGHIKLMNPQRSTVWY";
        static const uint8_t codes[20] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 18, 19, 20, 21};
        for (int i = 0; i < 20; i++)
            if (uint8_t(letters[i]) == c) return codes[i];
        return 0;
    }
};

/// SequenceAA: one byte per residue, un-encoded (kmeraa.rs:404-484)
class SequenceAA {
  public:
    explicit SequenceAA(std::string_view s) : seq_(s.begin(), s.end()) {}
    SequenceAA(const uint8_t *raw, size_t n) : seq_(raw, raw + n) {}
    static SequenceAA from_str(std::string_view s) { return SequenceAA(s); }
    /// drops the bytes that are not amino-acid letters (kmeraa.rs:447-456)
    static SequenceAA new_filtered(const uint8_t *raw, size_t n) {
        SequenceAA s(std::string_view{});
        for (size_t i = 0; i < n; i++)
            if (AlphabetAA::is_valid_base(raw[i])) s.seq_.push_back(raw[i]);
        return s;
    }
    size_t len() const { return seq_.size(); }
    size_t size() const { return seq_.size(); }
    const std::vector<uint8_t> &bytes() const { return seq_; }

  private:
    std::vector<uint8_t> seq_;
};

template <class V, int TYPE, int KMAX> struct KmerAAbits {
    using Val = V;
    static constexpr int kmu_type = TYPE;
    static constexpr bool is_aa = true;
    V v = 0;
    uint8_t k = 0;

    KmerAAbits() = default;
    KmerAAbits(V val, uint8_t nb) : v(val), k(nb) {}
    static KmerAAbits build(V val, uint8_t nb) { return KmerAAbits(val, nb); }
    static KmerAAbits from_raw(uint64_t raw, uint8_t nb) { return KmerAAbits(V(raw), nb); }
    static constexpr size_t get_nb_base_max() { return KMAX; }
    static constexpr size_t get_bitsize() { return 8 * sizeof(V); }
    uint8_t get_nb_base() const { return k; }
    V get_compressed_value() const { return v; }
    V raw() const { return v; }
    /// `c` is the residue letter: push encodes (kmeraa.rs:303-306)
    KmerAAbits push(uint8_t c) const {
        return KmerAAbits(V(((v << 5) & ((V(1) << (5 * k)) - 1)) | AlphabetAA::encode(c)), k);
    }
    KmerAAbits reverse_complement() const { throw std::logic_error("KmerAA: reverse_complement makes no sense"); }  // :315
    bool operator<(const KmerAAbits &o) const { return k != o.k ? k < o.k : v < o.v; }
    bool operator==(const KmerAAbits &o) const { return v == o.v && k == o.k; }
};
using KmerAA32bit = KmerAAbits<uint32_t, KMU_KMERAA32BIT, 6>;    // kmeraa.rs:146-274
using KmerAA64bit = KmerAAbits<uint64_t, KMU_KMERAA64BIT, 12>;   // kmeraa.rs:280-397

// =====================================================================================================================
// fhash: the closures of the reference's callers, by name
// =====================================================================================================================

enum class FHash : int {
    identity_raw = KMU_FHASH_IDENTITY_RAW,       // |kmer| kmer.0
    value_masked = KMU_FHASH_VALUE_MASKED,       // |kmer| kmer.get_compressed_value() & mask
    canon_raw = KMU_FHASH_CANON_RAW,             // |kmer| kmer.reverse_complement().min(*kmer).0
    canon_invhash = KMU_FHASH_CANON_INVHASH,     // |kmer| intNN_hash(kmer.reverse_complement().min(*kmer).0)
    invhash_raw = KMU_FHASH_INVHASH_RAW,         // |kmer| intNN_hash(kmer.0)
    canon_value = KMU_FHASH_CANON_VALUE,         // |kmer| min(kmer, revcomp).get_compressed_value()
    canon_nthash = KMU_FHASH_CANON_NTHASH,
    canon_nthash_8b = KMU_FHASH_CANON_NTHASH_8B,
};
/// the names the reference's tests give these closures (seqsketchjaccard.rs:768-775, aautils/setsketchert.rs:1235-1241)
inline constexpr FHash kmer_revcomp_hash_fn = FHash::canon_invhash;
inline constexpr FHash kmer_identity = FHash::identity_raw;
inline constexpr FHash kmer_hash_fn = FHash::value_masked;

// =====================================================================================================================
// gathering sequences into the (bytes, offsets) form of the C-ABI
// =====================================================================================================================

namespace detail {

struct Batch {
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> offsets;         // in bases / residues, n + 1 (always on the host: sizes, block layouts)
    std::vector<uint64_t> packed_offsets;  // byte offset of every packed sequence (DNA)
    int input_kind = KMU_INPUT_ASCII;
    // device-resident batch (reads left on the GPU by the FASTX reader): these pointers are used instead of `bytes` and
    // the host `offsets`; the offsets index the whole device array, so a range of reads needs no copy
    const uint8_t *dev_bytes = nullptr;
    const uint64_t *dev_offsets = nullptr;
    bool on_device() const { return dev_bytes != nullptr; }
    int mem() const { return on_device() ? KMU_MEM_DEVICE : KMU_MEM_HOST; }
    const uint8_t *bytes_ptr() const { return on_device() ? dev_bytes : bytes.data(); }
    const uint64_t *offsets_ptr() const { return on_device() ? dev_offsets : offsets.data(); }
    uint32_t n() const { return uint32_t(offsets.size() - 1); }
    const uint64_t *packed_ptr() const { return input_kind == KMU_INPUT_PACKED2 ? packed_offsets.data() : nullptr; }
};

inline Batch gather(const std::vector<const Sequence *> &vseq) {
    Batch b;
    b.input_kind = KMU_INPUT_PACKED2;
    b.offsets.assign(1, 0);
    for (const Sequence *s : vseq) {
        b.packed_offsets.push_back(b.bytes.size());
        b.bytes.insert(b.bytes.end(), s->packed().begin(), s->packed().end());
        b.offsets.push_back(b.offsets.back() + s->size());
    }
    if (b.packed_offsets.empty()) b.packed_offsets.push_back(0);
    b.bytes.resize(b.bytes.size() + 16, 0);
    return b;
}

inline Batch gather(const std::vector<const SequenceAA *> &vseq) {
    Batch b;
    b.offsets.assign(1, 0);
    for (const SequenceAA *s : vseq) {
        b.bytes.insert(b.bytes.end(), s->bytes().begin(), s->bytes().end());
        b.offsets.push_back(b.offsets.back() + s->len());
    }
    b.bytes.resize(b.bytes.size() + 16, 0);
    return b;
}

/// a packed (Sequence) batch as ASCII, for the entry points that take unpacked bases only
inline Batch unpacked(const Batch &b) {
    if (b.input_kind != KMU_INPUT_PACKED2) return b;
    Batch a;
    a.offsets = b.offsets;
    a.bytes.resize(b.offsets.back() + 16, 0);
    for (uint32_t i = 0; i < b.n(); i++)
        for (uint64_t p = 0, len = b.offsets[i + 1] - b.offsets[i]; p < len; p++)
            a.bytes[b.offsets[i] + p] = Alphabet2b::decode(uint8_t(b.bytes[b.packed_offsets[i] + (p >> 2)] >> (6 - 2 * (p & 3))));
    return a;
}

template <class Seq> std::vector<const Seq *> pointers(const std::vector<Seq> &v) {
    std::vector<const Seq *> p;
    p.reserve(v.size());
    for (const Seq &s : v) p.push_back(&s);
    return p;
}

template <class Sig> constexpr int sig_type_of() {
    if constexpr (std::is_same_v<Sig, uint32_t>) return KMU_SIG_U32;
    else if constexpr (std::is_same_v<Sig, uint64_t>) return KMU_SIG_U64;
    else if constexpr (std::is_same_v<Sig, float>) return KMU_SIG_F32;
    else {
        static_assert(std::is_same_v<Sig, double>, "signature element must be u32, u64, f32 or f64");
        return KMU_SIG_F64;
    }
}

template <class Sig> std::vector<std::vector<Sig>> split_rows(const std::vector<Sig> &flat, size_t rows, size_t m) {
    std::vector<std::vector<Sig>> out(rows);
    for (size_t r = 0; r < rows; r++) out[r].assign(flat.begin() + r * m, flat.begin() + (r + 1) * m);
    return out;
}

inline kmu_sketch_params sketch_params(int algo, int kmer_type, size_t k, size_t m, int sig_type, int hasher, int fhash,
                                       int block_size, int mode, int input_kind) {
    kmu_sketch_params p{};
    p.algo = algo;
    p.kmer_type = kmer_type;
    p.kmer_size = int32_t(k);
    p.sketch_size = int32_t(m);
    p.sig_type = sig_type;
    p.hasher = hasher;
    p.fhash = fhash;
    p.block_size = block_size;
    p.mode = mode;
    p.input_kind = input_kind;
    p.mem = KMU_MEM_HOST;
    return p;
}

/// raw `.0` of every k-mer of every sequence, generated on the device (KmerSeqIterator::next, kmergenerator.rs:75-106)
template <class Kmer> std::vector<std::vector<Kmer>> device_kmers(Context &ctx, const Batch &b, size_t k) {
    if (b.on_device()) throw std::invalid_argument("k-mer lists / host closures need the sequences in host memory");
    kmu_hash_params hp{};
    hp.kmer_type = Kmer::kmu_type;
    hp.kmer_size = int32_t(k);
    hp.fhash = KMU_FHASH_IDENTITY_RAW;
    hp.input_kind = b.input_kind;
    hp.mem = KMU_MEM_HOST;
    std::vector<uint64_t> raw(std::max<uint64_t>(b.offsets.back(), 1), 0);
    ctx.check(kmu_kmer_hashes(ctx.raw(), &hp, b.bytes.data(), b.offsets.data(), b.packed_ptr(), b.n(), raw.data()));
    std::vector<std::vector<Kmer>> out(b.n());
    for (uint32_t i = 0; i < b.n(); i++) {
        const uint64_t len = b.offsets[i + 1] - b.offsets[i];
        if (len < k) continue;   // a sequence shorter than k yields nothing (kmergenerator.rs:98-99)
        out[i].reserve(len - k + 1);
        for (uint64_t p = 0; p + k <= len; p++) out[i].push_back(Kmer::from_raw(raw[b.offsets[i] + p], uint8_t(k)));
    }
    return out;
}

/// rows of signatures for a batch, row-major in `flat` (resized to rows * m; a caller that sketches pack after pack keeps
/// one such buffer), either with a device-side closure (FHash) or a host callable.  Returns the number of rows.
template <class Kmer, class Sig, class F>
size_t run_sketch_flat(Context &ctx, const Batch &b, int algo, size_t k, size_t m, int hasher, F fhash, int mode,
                       std::vector<Sig> &flat, uint32_t flags = 0) {
    const size_t rows = mode == KMU_MODE_ALL_SEQS ? 1 : b.n();
    flat.resize(std::max<size_t>(rows, 1) * m);
    if constexpr (std::is_same_v<F, FHash>) {
        kmu_sketch_params p =
            sketch_params(algo, Kmer::kmu_type, k, m, sig_type_of<Sig>(), hasher, int(fhash), 0, mode, b.input_kind);
        p.flags = flags;
        p.mem = b.mem();
        if (b.on_device()) {   // reads already on the GPU: only the signatures cross PCIe
            DeviceBuffer d_sig(ctx, flat.size() * sizeof(Sig));
            ctx.check(kmu_sketch(ctx.raw(), &p, b.dev_bytes, b.dev_offsets, nullptr, b.n(), nullptr, d_sig.data(), nullptr));
            d_sig.download(flat.data(), flat.size() * sizeof(Sig));
        } else {
            ctx.check(kmu_sketch(ctx.raw(), &p, b.bytes.data(), b.offsets.data(), b.packed_ptr(), b.n(), nullptr, flat.data(),
                                 nullptr));
        }
    } else {
        static_assert(std::is_invocable_r_v<typename Kmer::Val, F, const Kmer &>, "fhash must be Val(const Kmer&)");
        using Val = typename Kmer::Val;
        auto kmers = device_kmers<Kmer>(ctx, b, k);
        std::vector<Val> hashed;
        std::vector<uint64_t> off(1, 0);
        for (const auto &seq : kmers) {
            for (const Kmer &km : seq) hashed.push_back(fhash(km));
            off.push_back(hashed.size());
        }
        if (hashed.empty()) hashed.push_back(0);
        kmu_sketch_params p = sketch_params(algo, Kmer::kmu_type, k, m, sig_type_of<Sig>(), hasher, KMU_FHASH_IDENTITY_RAW, 0,
                                            mode, KMU_INPUT_ASCII);
        p.flags = flags;
        ctx.check(kmu_sketch_hashed(ctx.raw(), &p, hashed.data(), off.data(), b.n(), flat.data(), nullptr));
    }
    return rows;
}

template <class Kmer, class Sig, class F>
std::vector<std::vector<Sig>> run_sketch(Context &ctx, const Batch &b, int algo, size_t k, size_t m, int hasher, F fhash,
                                         int mode, uint32_t flags = 0) {
    std::vector<Sig> flat;
    const size_t rows = run_sketch_flat<Kmer, Sig>(ctx, b, algo, k, m, hasher, fhash, mode, flat, flags);
    return split_rows(flat, rows, m);
}

}  // namespace detail

// =====================================================================================================================
// base: KmerGenerator
// =====================================================================================================================

/// KmerGenerator::new(k).generate_kmer(&seq) -> Vec<Kmer>: all L - k + 1 forward k-mers (kmergenerator.rs:117-239)
template <class Kmer> class KmerGenerator {
  public:
    explicit KmerGenerator(uint8_t kmer_size, Context &ctx = Context::global()) : k_(kmer_size), ctx_(ctx) {}
    std::vector<Kmer> generate_kmer(const Sequence &seq) const {
        return std::move(detail::device_kmers<Kmer>(ctx_, detail::gather(std::vector<const Sequence *>{&seq}), k_)[0]);
    }
    std::vector<Kmer> generate_kmer(const SequenceAA &seq) const {
        return std::move(detail::device_kmers<Kmer>(ctx_, detail::gather(std::vector<const SequenceAA *>{&seq}), k_)[0]);
    }
    /// generate_kmer_in_range(&seq, begin, end) (kmergenerator.rs:169-174; kmeraa.rs:667-672): the k-mers inside bases
    /// [begin, end).  A bad range throws KmuError(KMU_E_BAD_ARG) where the reference unwraps set_range's Err.
    template <class Seq> std::vector<Kmer> generate_kmer_in_range(const Seq &seq, size_t begin, size_t end) const {
        if (begin >= end) throw KmuError(KMU_E_BAD_ARG, "KmerGenerationPattern bad range for kmer iteration");  // kmergenerator.rs:284-286
        const detail::Batch b = detail::gather(std::vector<const Seq *>{&seq});
        kmu_hash_params hp{};
        hp.kmer_type = Kmer::kmu_type;
        hp.kmer_size = int32_t(k_);
        hp.fhash = KMU_FHASH_IDENTITY_RAW;
        hp.input_kind = b.input_kind;
        hp.mem = KMU_MEM_HOST;
        std::vector<uint64_t> raw(std::max<uint64_t>(b.offsets.back(), 1), 0);
        const uint64_t rb = begin, re = end;
        ctx_.check(kmu_kmer_hashes_range(ctx_.raw(), &hp, b.bytes.data(), b.offsets.data(), b.packed_ptr(), 1, &rb, &re, raw.data()));
        std::vector<Kmer> out;
        for (uint64_t p = begin; p + k_ <= end; p++) out.push_back(Kmer::from_raw(raw[p], k_));
        return out;
    }
    /// generate_weighted_kmer(&seq) -> FnvHashMap<Kmer, u32> (kmergenerator.rs:176-181): distinct k-mers with multiplicities,
    /// counted on the device (kmu_kmer_distribution); returned as (k-mer, count) pairs in no particular order
    template <class Seq> std::vector<std::pair<Kmer, uint32_t>> generate_weighted_kmer(const Seq &seq) const {
        const detail::Batch b = detail::gather(std::vector<const Seq *>{&seq});
        kmu_hash_params hp{};
        hp.kmer_type = Kmer::kmu_type;
        hp.kmer_size = int32_t(k_);
        hp.fhash = KMU_FHASH_IDENTITY_RAW;
        hp.input_kind = b.input_kind;
        hp.mem = KMU_MEM_HOST;
        uint64_t n = 0;
        ctx_.check(kmu_kmer_distribution(ctx_.raw(), &hp, b.bytes.data(), b.offsets.data(), b.packed_ptr(), 1, nullptr, nullptr, 0,
                                         nullptr, &n));
        std::vector<uint64_t> kk(std::max<uint64_t>(n, 1));
        std::vector<uint32_t> cc(std::max<uint64_t>(n, 1));
        ctx_.check(kmu_kmer_distribution(ctx_.raw(), &hp, b.bytes.data(), b.offsets.data(), b.packed_ptr(), 1, kk.data(), cc.data(), n,
                                         nullptr, &n));
        std::vector<std::pair<Kmer, uint32_t>> out;
        out.reserve(n);
        for (uint64_t i = 0; i < n; i++) out.emplace_back(Kmer::from_raw(kk[i], k_), cc[i]);
        return out;
    }

  private:
    uint8_t k_;
    Context &ctx_;
};

// =====================================================================================================================
// sketching: parameters and the trait
// =====================================================================================================================

enum class DataType { DNA, AA };                                              // sketcharg.rs:13-16
enum class SketchAlgo { PROB3A, SUPER, SUPER2, OPTDENS, REVOPTDENS, HLL };    // sketcharg.rs:26-33

namespace detail {
inline const char *const ALGO_NAMES[6] = {"PROB3A", "SUPER", "SUPER2", "OPTDENS", "REVOPTDENS", "HLL"};
inline const char *const DATA_NAMES[2] = {"DNA", "AA"};
/// value of "key" in a flat JSON object written by serde_json (no nesting, no escapes in what these structs hold)
inline std::string json_field(const std::string &text, const std::string &key) {
    const std::string tag = "\"" + key + "\":";
    const size_t at = text.find(tag);
    if (at == std::string::npos) throw std::runtime_error("missing field " + key);
    size_t b = at + tag.size(), e = b;
    while (e < text.size() && text[e] != ',' && text[e] != '}') e++;
    std::string v = text.substr(b, e - b);
    if (v.size() >= 2 && v.front() == '"') v = v.substr(1, v.size() - 2);
    return v;
}
inline std::string read_text_file(const std::string &path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("reload_json could not open file " + path);
    return std::string((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
}  // namespace detail

class SeqSketcherParams {   // sketcharg.rs:40-138
  public:
    SeqSketcherParams(size_t kmer_size, size_t sketch_size, SketchAlgo algo, DataType data_t)
        : kmer_size_(kmer_size), sketch_size_(sketch_size), algo_(algo), data_t_(data_t) {}
    size_t get_kmer_size() const { return kmer_size_; }
    size_t get_sketch_size() const { return sketch_size_; }
    SketchAlgo get_algo() const { return algo_; }
    DataType get_data_t() const { return data_t_; }
    /// dump_json: serde_json::to_writer(&self) -- {"kmer_size":8,"sketch_size":200,"algo":"PROB3A","data_t":"DNA"}
    void dump_json(const std::string &filename) const {
        std::ofstream out(filename);
        if (!out) throw std::runtime_error("SeqSketcher dump failed");
        out << "{\"kmer_size\":" << kmer_size_ << ",\"sketch_size\":" << sketch_size_ << ",\"algo\":\""
            << detail::ALGO_NAMES[int(algo_)] << "\",\"data_t\":\"" << detail::DATA_NAMES[int(data_t_)] << "\"}";
    }
    /// reload_json(dirpath): <dirpath>/sketchparams_dump.json
    static SeqSketcherParams reload_json(const std::string &dirpath) {
        const std::string text = detail::read_text_file(dirpath + "/sketchparams_dump.json");
        const std::string algo = detail::json_field(text, "algo"), data_t = detail::json_field(text, "data_t");
        int ai = -1, di = -1;
        for (int i = 0; i < 6; i++) if (algo == detail::ALGO_NAMES[i]) ai = i;
        for (int i = 0; i < 2; i++) if (data_t == detail::DATA_NAMES[i]) di = i;
        if (ai < 0 || di < 0) throw std::runtime_error("reload_json: unknown algo / data_t");
        return SeqSketcherParams(std::stoul(detail::json_field(text, "kmer_size")), std::stoul(detail::json_field(text, "sketch_size")),
                                 SketchAlgo(ai), DataType(di));
    }

  private:
    size_t kmer_size_, sketch_size_;
    SketchAlgo algo_;
    DataType data_t_;
};

/// trait SeqSketcherT<Kmer> (setsketchert.rs:54-80) and SeqSketcherAAT (aautils/setsketchert.rs:42-72) in one: the
/// sequence type follows the k-mer type.  `Sig` is the trait's associated type.
template <class Kmer, class SigT> class SeqSketcherT {
  public:
    using Sig = SigT;
    using Seq = std::conditional_t<Kmer::is_aa, SequenceAA, Sequence>;
    virtual ~SeqSketcherT() = default;
    virtual size_t get_kmer_size() const = 0;
    virtual size_t get_sketch_size() const = 0;
    virtual SketchAlgo get_algo() const = 0;
    /// one signature per sequence, row i <-> vseq[i]
    virtual std::vector<std::vector<Sig>> sketch_compressedkmer(const std::vector<const Seq *> &vseq, FHash fhash) const = 0;
    /// ONE signature for the whole list (outer length 1)
    virtual std::vector<std::vector<Sig>> sketch_compressedkmer_seqs(const std::vector<const Seq *> &vseq,
                                                                     FHash fhash) const = 0;
    // the AA trait's names for the same two calls
    std::vector<std::vector<Sig>> sketch_compressedkmeraa(const std::vector<const Seq *> &vseq, FHash fhash) const {
        return sketch_compressedkmer(vseq, fhash);
    }
    std::vector<std::vector<Sig>> sketch_compressedkmeraa_seqs(const std::vector<const Seq *> &vseq, FHash fhash) const {
        return sketch_compressedkmer_seqs(vseq, fhash);
    }
};

namespace detail {
/// shared body of the three implementers
template <class Kmer, class Sig, int ALGO> class SketcherImpl : public SeqSketcherT<Kmer, Sig> {
  public:
    using Seq = typename SeqSketcherT<Kmer, Sig>::Seq;
    SketcherImpl(const SeqSketcherParams &params, SketchAlgo algo, int hasher, Context &ctx)
        : params_(params), algo_(algo), hasher_(hasher), ctx_(ctx) {}
    size_t get_kmer_size() const override { return params_.get_kmer_size(); }
    size_t get_sketch_size() const override { return params_.get_sketch_size(); }
    SketchAlgo get_algo() const override { return algo_; }
    std::vector<std::vector<Sig>> sketch_compressedkmer(const std::vector<const Seq *> &vseq, FHash fhash) const override {
        return run(vseq, fhash, KMU_MODE_PER_SEQ);
    }
    std::vector<std::vector<Sig>> sketch_compressedkmer_seqs(const std::vector<const Seq *> &vseq,
                                                             FHash fhash) const override {
        return run(vseq, fhash, KMU_MODE_ALL_SEQS);
    }
    /// arbitrary closure: evaluated on the host, multiset + sketch on the device
    template <class F, class = std::enable_if_t<!std::is_same_v<std::decay_t<F>, FHash>>>
    std::vector<std::vector<Sig>> sketch_compressedkmer(const std::vector<const Seq *> &vseq, F fhash) const {
        return run(vseq, fhash, KMU_MODE_PER_SEQ);
    }
    template <class F, class = std::enable_if_t<!std::is_same_v<std::decay_t<F>, FHash>>>
    std::vector<std::vector<Sig>> sketch_compressedkmer_seqs(const std::vector<const Seq *> &vseq, F fhash) const {
        return run(vseq, fhash, KMU_MODE_ALL_SEQS);
    }

  private:
    template <class F> std::vector<std::vector<Sig>> run(const std::vector<const Seq *> &vseq, F fhash, int mode) const {
        return run_sketch<Kmer, Sig>(ctx_, gather(vseq), ALGO, params_.get_kmer_size(), params_.get_sketch_size(), hasher_,
                                     fhash, mode);
    }
    SeqSketcherParams params_;
    SketchAlgo algo_;
    int hasher_;
    Context &ctx_;
};
}  // namespace detail

/// ProbHash3aSketch<Kmer>: type Sig = Kmer::Val (setsketchert.rs:85-203)
template <class Kmer> class ProbHash3aSketch : public detail::SketcherImpl<Kmer, typename Kmer::Val, KMU_ALGO_PROB3A> {
  public:
    explicit ProbHash3aSketch(const SeqSketcherParams &params, Context &ctx = Context::global())
        : detail::SketcherImpl<Kmer, typename Kmer::Val, KMU_ALGO_PROB3A>(params, SketchAlgo::PROB3A, KMU_HASHER_NOHASH, ctx) {}
};

/// SuperHashSketch<Kmer, S>: S = f32 / f64, NoHashHasher (setsketchert.rs:211-336)
template <class Kmer, class S> class SuperHashSketch : public detail::SketcherImpl<Kmer, S, KMU_ALGO_SUPER> {
    static_assert(std::is_floating_point_v<S>, "SuperHashSketch: S is f32 or f64");

  public:
    explicit SuperHashSketch(const SeqSketcherParams &params, Context &ctx = Context::global())
        : detail::SketcherImpl<Kmer, S, KMU_ALGO_SUPER>(params, SketchAlgo::SUPER, KMU_HASHER_NOHASH, ctx) {}
};

/// OptDensHashSketch<Kmer, S>: one-permutation hashing + optimal densification, S = f32 / f64 (setsketchert.rs:343-463;
/// AA: aautils/setsketchert.rs:482-612)
template <class Kmer, class S> class OptDensHashSketch : public detail::SketcherImpl<Kmer, S, KMU_ALGO_OPTDENS> {
    static_assert(std::is_floating_point_v<S>, "OptDensHashSketch: S is f32 or f64");

  public:
    explicit OptDensHashSketch(const SeqSketcherParams &params, Context &ctx = Context::global())
        : detail::SketcherImpl<Kmer, S, KMU_ALGO_OPTDENS>(params, SketchAlgo::OPTDENS, KMU_HASHER_NOHASH, ctx) {}
};

/// RevOptDensHashSketch<Kmer, S>: reverse optimal densification, robust when the sketch is larger than the sequence
/// (setsketchert.rs:474-599; AA: aautils/setsketchert.rs:616-746)
template <class Kmer, class S> class RevOptDensHashSketch : public detail::SketcherImpl<Kmer, S, KMU_ALGO_REVOPTDENS> {
    static_assert(std::is_floating_point_v<S>, "RevOptDensHashSketch: S is f32 or f64");

  public:
    explicit RevOptDensHashSketch(const SeqSketcherParams &params, Context &ctx = Context::global())
        : detail::SketcherImpl<Kmer, S, KMU_ALGO_REVOPTDENS>(params, SketchAlgo::REVOPTDENS, KMU_HASHER_NOHASH, ctx) {}
};

/// probminhash::setsketcher::SetSketchParams as HyperLogLogSketch::new takes it (defaults of the crate, as recalled)
struct SetSketchParams {
    double b = 1.001;
    size_t m = 4096;
    double a = 20.;
    uint32_t q = 65534;
};
/// HllSeqsThreading (setsketchert.rs:600-636): how upstream splits a list of sequences over threads; kept for the signature
struct HllSeqsThreading {
    size_t nb_iter_thread = 4, thread_threshold = 10000000;
    size_t get_nb_iter_threads() const { return nb_iter_thread; }
    size_t get_thread_threshold() const { return thread_threshold; }
};

/// HyperLogLogSketch<Kmer, S>::new(&seq_params, hll_params, hll_threads) (setsketchert.rs:640-896; AA:
/// aautils/setsketchert.rs:780-1011): SetSketch registers, S = u16 / u32 / u64; the sketch has hll_params.m registers
template <class Kmer, class S> class HyperLogLogSketch : public SeqSketcherT<Kmer, S> {
    static_assert(std::is_same_v<S, uint16_t> || std::is_same_v<S, uint32_t> || std::is_same_v<S, uint64_t>, "S is u16, u32 or u64");

  public:
    using Seq = typename SeqSketcherT<Kmer, S>::Seq;
    HyperLogLogSketch(const SeqSketcherParams &seq_params, SetSketchParams hll_params, HllSeqsThreading hll_threads = {},
                      Context &ctx = Context::global())
        : params_(seq_params), hll_(hll_params), threads_(hll_threads), ctx_(ctx) {}
    size_t get_kmer_size() const override { return params_.get_kmer_size(); }
    size_t get_sketch_size() const override { return params_.get_sketch_size(); }
    SketchAlgo get_algo() const override { return SketchAlgo::HLL; }
    std::vector<std::vector<S>> sketch_compressedkmer(const std::vector<const Seq *> &vseq, FHash fhash) const override {
        return run(vseq, fhash, KMU_MODE_PER_SEQ);
    }
    std::vector<std::vector<S>> sketch_compressedkmer_seqs(const std::vector<const Seq *> &vseq, FHash fhash) const override {
        return run(vseq, fhash, KMU_MODE_ALL_SEQS);
    }

  private:
    std::vector<std::vector<S>> run(const std::vector<const Seq *> &vseq, FHash fhash, int mode) const {
        kmu_hll_params hp{hll_.b, hll_.a, hll_.q, 0};
        ctx_.check(kmu_set_hll_params(ctx_.raw(), &hp));
        const detail::Batch b = detail::gather(vseq);
        const int sig = std::is_same_v<S, uint16_t> ? KMU_SIG_U16 : std::is_same_v<S, uint32_t> ? KMU_SIG_U32 : KMU_SIG_U64;
        kmu_sketch_params p = detail::sketch_params(KMU_ALGO_HLL, Kmer::kmu_type, params_.get_kmer_size(), hll_.m, sig,
                                                    KMU_HASHER_NOHASH, int(fhash), 0, mode, b.input_kind);
        const size_t rows = mode == KMU_MODE_ALL_SEQS ? 1 : b.n();
        std::vector<S> flat(std::max<size_t>(rows, 1) * hll_.m);
        ctx_.check(kmu_sketch(ctx_.raw(), &p, b.bytes.data(), b.offsets.data(), b.packed_ptr(), b.n(), nullptr, flat.data(), nullptr));
        return detail::split_rows(flat, rows, hll_.m);
    }
    SeqSketcherParams params_;
    SetSketchParams hll_;
    HllSeqsThreading threads_;
    Context &ctx_;
};

/// the `H: Hasher` parameter of SuperHash2Sketch
struct NoHashHasher { static constexpr int id = KMU_HASHER_NOHASH; };   // src/nohasher.rs
struct FnvHasher { static constexpr int id = KMU_HASHER_FNV1A; };       // fnv crate

/// SuperHash2Sketch<Kmer, S, H>: S = u32 / u64 (setsketchert.rs:904-1046)
template <class Kmer, class S, class H = NoHashHasher>
class SuperHash2Sketch : public detail::SketcherImpl<Kmer, S, KMU_ALGO_SUPER2> {
    static_assert(std::is_same_v<S, uint32_t> || std::is_same_v<S, uint64_t>, "SuperHash2Sketch: S is u32 or u64");

  public:
    explicit SuperHash2Sketch(const SeqSketcherParams &params, Context &ctx = Context::global())
        : detail::SketcherImpl<Kmer, S, KMU_ALGO_SUPER2>(params, SketchAlgo::SUPER2, H::id, ctx) {}
};

// =====================================================================================================================
// sketching: the older struct API, SeqSketcher (seqsketchjaccard.rs:117-415)
// =====================================================================================================================

class SeqSketcher {
  public:
    SeqSketcher(size_t kmer_size, size_t sketch_size, Context &ctx = Context::global())
        : kmer_size_(kmer_size), sketch_size_(sketch_size), ctx_(ctx) {}
    size_t get_kmer_size() const { return kmer_size_; }
    size_t get_sketch_size() const { return sketch_size_; }

    /// seqsketchjaccard.rs:211-260 (DNA), aautils/setsketchert.rs:114-150 (AA): Vec<Vec<Kmer::Val>>, row i <-> sequence i
    template <class Kmer, class Seq, class F>
    std::vector<std::vector<typename Kmer::Val>> sketch_probminhash3a(const std::vector<const Seq *> &vseq, F fhash) const {
        return detail::run_sketch<Kmer, typename Kmer::Val>(ctx_, detail::gather(vseq), KMU_ALGO_PROB3A, kmer_size_,
                                                            sketch_size_, KMU_HASHER_NOHASH, fhash, KMU_MODE_PER_SEQ);
    }
    /// the same on reads already gathered in the C-ABI form (what the FASTQ reader below returns)
    template <class Kmer, class F>
    std::vector<std::vector<typename Kmer::Val>> sketch_probminhash3a(const detail::Batch &reads, F fhash) const {
        return detail::run_sketch<Kmer, typename Kmer::Val>(ctx_, reads, KMU_ALGO_PROB3A, kmer_size_, sketch_size_,
                                                            KMU_HASHER_NOHASH, fhash, KMU_MODE_PER_SEQ);
    }
    /// ... and with the rows left row-major in a buffer the caller keeps from pack to pack (the tools' loop)
    template <class Kmer, class F>
    size_t sketch_probminhash3a(const detail::Batch &reads, F fhash, std::vector<typename Kmer::Val> &rows_out) const {
        return detail::run_sketch_flat<Kmer, typename Kmer::Val>(ctx_, reads, KMU_ALGO_PROB3A, kmer_size_, sketch_size_,
                                                                 KMU_HASHER_NOHASH, fhash, KMU_MODE_PER_SEQ, rows_out);
    }
    /// seqsketchjaccard.rs:272-319: ProbMinHash3 over the same weighted multiset
    template <class Kmer, class Seq, class F>
    std::vector<std::vector<typename Kmer::Val>> sketch_probminhash3(const std::vector<const Seq *> &vseq, F fhash) const {
        return detail::run_sketch<Kmer, typename Kmer::Val>(ctx_, detail::gather(vseq), KMU_ALGO_PROB3, kmer_size_,
                                                            sketch_size_, KMU_HASHER_NOHASH, fhash, KMU_MODE_PER_SEQ);
    }
    /// seqsketchjaccard.rs:328-380: SuperMinHash<S, Kmer::Val, fnv::FnvHasher>
    template <class Kmer, class S = double, class Seq, class F>
    std::vector<std::vector<S>> sketch_superminhash(const std::vector<const Seq *> &vseq, F fhash) const {
        return detail::run_sketch<Kmer, S>(ctx_, detail::gather(vseq), KMU_ALGO_SUPER, kmer_size_, sketch_size_,
                                           KMU_HASHER_FNV1A, fhash, KMU_MODE_PER_SEQ);
    }

    /// dump_json / reload_json (seqsketchjaccard.rs:142-201): {"kmer_size":..,"sketch_size":..}
    void dump_json(const std::string &filename) const {
        std::ofstream out(filename);
        if (!out) throw std::runtime_error("SeqSketcher dump failed");
        out << "{\"kmer_size\":" << kmer_size_ << ",\"sketch_size\":" << sketch_size_ << "}";
    }
    static SeqSketcher reload_json(const std::string &dirpath, Context &ctx = Context::global()) {
        const std::string text = detail::read_text_file(dirpath + "/sketchparams_dump.json");
        return SeqSketcher(std::stoul(detail::json_field(text, "kmer_size")), std::stoul(detail::json_field(text, "sketch_size")), ctx);
    }
    /// create_signature_dump: magic, sig_size = 4, sketch_size, kmer_size as four u32 (seqsketchjaccard.rs:385-414)
    std::ofstream create_signature_dump(const std::string &dumpfname) const {
        std::ofstream out(dumpfname, std::ios::binary);
        if (!out) throw std::runtime_error("create_signature_dump: cannot open " + dumpfname);
        const uint32_t head[4] = {MAGIC_SIG_DUMP, 4u, uint32_t(sketch_size_), uint32_t(kmer_size_)};
        out.write(reinterpret_cast<const char *>(head), sizeof head);
        return out;
    }
    /// dump_signatures_block_u32 (seqsketchjaccard.rs:572-583)
    static void dump_signatures_block_u32(const std::vector<std::vector<uint32_t>> &signatures, std::ostream &out) {
        for (const auto &sig : signatures) out.write(reinterpret_cast<const char *>(sig.data()), std::streamsize(4 * sig.size()));
    }
    static void dump_signatures_block_u32(const std::vector<uint32_t> &rows, size_t n_values, std::ostream &out) {
        out.write(reinterpret_cast<const char *>(rows.data()), std::streamsize(4 * n_values));
    }
    static constexpr uint32_t MAGIC_SIG_DUMP = 0xceabeadd;   // seqsketchjaccard.rs:570

  private:
    size_t kmer_size_, sketch_size_;
    Context &ctx_;
};

/// SigSketchFileReader::new / next (seqsketchjaccard.rs:586-712).  Upstream's `next` reads the bytes of a row but returns an
/// empty Vec (:690-708); this reader returns the row.
class SigSketchFileReader {
  public:
    explicit SigSketchFileReader(const std::string &fname) : in_(fname, std::ios::binary) {
        uint32_t head[4] = {0, 0, 0, 0};
        if (!in_) throw std::runtime_error("SigSketchFileReader could not open file " + fname);
        in_.read(reinterpret_cast<char *>(head), 16);
        if (in_.gcount() < 4) throw std::runtime_error("SigSketchFileReader could no read magic");
        if (head[0] != SeqSketcher::MAGIC_SIG_DUMP) throw std::runtime_error("file is not a dump of signature");
        if (in_.gcount() < 16) throw std::runtime_error("SigSketchFileReader could no read sketch_size");
        sig_size_ = head[1];
        sketch_size_ = head[2];
        kmer_size_ = uint8_t(head[3]);
        if (sig_size_ != 4) throw std::runtime_error("SigSketchFileReader , sig_size != 4 not yet implemented");
    }
    uint8_t get_kmer_size() const { return kmer_size_; }
    size_t get_signature_length() const { return sketch_size_; }
    size_t get_signature_size() const { return sig_size_; }
    /// the next signature, or nothing at the end of the file
    std::optional<std::vector<uint32_t>> next() {
        std::vector<uint32_t> row(sketch_size_);
        in_.read(reinterpret_cast<char *>(row.data()), std::streamsize(4 * sketch_size_));
        if (size_t(in_.gcount()) < 4 * sketch_size_) return std::nullopt;
        return row;
    }

  private:
    std::ifstream in_;
    size_t sig_size_ = 0, sketch_size_ = 0;
    uint8_t kmer_size_ = 0;
};

// =====================================================================================================================
// what callers do with signatures (seqsketchjaccard.rs:58-108, 423-495)
// =====================================================================================================================

namespace detail {
template <class D> uint32_t equal_slots(Context &ctx, const std::vector<D> &a, const std::vector<D> &b) {
    if (a.size() != b.size()) throw std::invalid_argument("signatures of different lengths");
    static_assert(sizeof(D) == 4 || sizeof(D) == 8, "signature element of 4 or 8 bytes");
    const uint32_t zero = 0;
    uint32_t out = 0;
    ctx.check(kmu_sig_equal_pairs(ctx.raw(), a.data(), 1, b.data(), 1, uint32_t(a.size()), sig_type_of<D>(), &zero, &zero, 1,
                                  KMU_MEM_HOST, &out));
    return out;
}
}  // namespace detail

/// compute_probminhash_jaccard (crate probminhash::jaccard): fraction of equal slots
template <class D> double compute_probminhash_jaccard(const std::vector<D> &siga, const std::vector<D> &sigb,
                                                      Context &ctx = Context::global()) {
    return double(detail::equal_slots(ctx, siga, sigb)) / double(siga.size());
}
/// get_jaccard_index_estimate of SuperMinHash sketches: likewise (seqsketchjaccard.rs:727-732)
template <class S> double compute_superminhash_jaccard(const std::vector<S> &hsketch, const std::vector<S> &other,
                                                       Context &ctx = Context::global()) {
    return double(detail::equal_slots(ctx, hsketch, other)) / double(hsketch.size());
}
/// probminhash_get_jaccard_objects -> (jp, Some(common objects) | None), seqsketchjaccard.rs:86-108
template <class D>
std::pair<double, std::optional<std::vector<D>>> probminhash_get_jaccard_objects(const std::vector<D> &siga,
                                                                                 const std::vector<D> &sigb,
                                                                                 Context &ctx = Context::global()) {
    const uint32_t inter = detail::equal_slots(ctx, siga, sigb);
    if (inter == 0) return {0.0, std::nullopt};
    std::vector<D> common;
    for (size_t i = 0; i < siga.size(); i++)
        if (siga[i] == sigb[i]) common.push_back(siga[i]);
    return {double(inter) / double(siga.size()), std::move(common)};
}

/// jaccard_index_probminhash3a(seqa, vseqb, sketch_size, kmer_size, fhash) -> Vec<f64> (seqsketchjaccard.rs:423-495):
/// P-Jaccard estimate of seqa against every sequence of vseqb
template <class Kmer, class F>
std::vector<double> jaccard_index_probminhash3a(const Sequence &seqa, const std::vector<Sequence> &vseqb, size_t sketch_size,
                                                size_t kmer_size, F fhash, Context &ctx = Context::global()) {
    std::vector<const Sequence *> all{&seqa};
    for (const Sequence &s : vseqb) all.push_back(&s);
    auto sigs = SeqSketcher(kmer_size, sketch_size, ctx).sketch_probminhash3a<Kmer>(all, fhash);
    std::vector<double> jac;
    for (size_t i = 1; i < sigs.size(); i++) jac.push_back(compute_probminhash_jaccard(sigs[0], sigs[i], ctx));
    return jac;
}

// =====================================================================================================================
// block sketching (seqblocksketch.rs)
// =====================================================================================================================

struct BlockSketched {   // seqblocksketch.rs:38-66
    uint32_t numseq;
    uint32_t numblock;
    std::vector<uint32_t> sketch;
    void dump(std::ostream &out) const {
        out.write(reinterpret_cast<const char *>(&numseq), 4);
        out.write(reinterpret_cast<const char *>(&numblock), 4);
        out.write(reinterpret_cast<const char *>(sketch.data()), std::streamsize(4 * sketch.size()));
    }
};

struct BlockSketchedSeq {   // seqblocksketch.rs:73-76
    uint32_t numseq;
    std::vector<BlockSketched> sketch;
};

/// Distance<BlockSketched>: 1 between blocks of one sequence, else the fraction of differing slots (seqblocksketch.rs:419-431)
class DistBlockSketched {
  public:
    explicit DistBlockSketched(Context &ctx = Context::global()) : ctx_(ctx) {}
    float eval(const BlockSketched &va, const BlockSketched &vb) const {
        if (va.numseq == vb.numseq) return 1.f;
        const uint32_t eq = detail::equal_slots(ctx_, va.sketch, vb.sketch);
        return float(va.sketch.size() - eq) / float(va.sketch.size());
    }

  private:
    Context &ctx_;
};

/// BlockSeqSketcher{block_size, kmer_size, sketch_size}: Kmer32bit / u32 only (seqblocksketch.rs:79-227)
class BlockSeqSketcher {
  public:
    BlockSeqSketcher(size_t block_size, size_t kmer_size, size_t sketch_size, Context &ctx = Context::global())
        : block_size_(block_size), kmer_size_(kmer_size), sketch_size_(sketch_size), ctx_(ctx) {}

    /// blocksketch_sequence(numseq, seq, fhash): seqblocksketch.rs:97-149
    BlockSketchedSeq blocksketch_sequence(size_t numseq, const Sequence &seq, FHash fhash) const {
        return std::move(blocksketch_sequences(numseq, {&seq}, fhash)[0]);
    }
    /// blocksketch_sequences(numfirst, seqs, fhash): sequence i gets numseq = numfirst + i (seqblocksketch.rs:152-167)
    std::vector<BlockSketchedSeq> blocksketch_sequences(size_t numfirst, const std::vector<const Sequence *> &seqs,
                                                        FHash fhash) const {
        return blocksketch_sequences(numfirst, detail::gather(seqs), fhash);
    }
    std::vector<BlockSketchedSeq> blocksketch_sequences(size_t numfirst, const detail::Batch &b, FHash fhash) const {
        std::vector<uint64_t> rows(b.n() + 1);
        ctx_.check(kmu_block_layout(b.offsets.data(), b.n(), uint32_t(block_size_), rows.data()));
        kmu_sketch_params p = detail::sketch_params(KMU_ALGO_PROB3A, KMU_KMER32BIT, kmer_size_, sketch_size_, KMU_SIG_U32,
                                                    KMU_HASHER_NOHASH, int(fhash), int(block_size_), KMU_MODE_PER_SEQ,
                                                    b.input_kind);
        p.mem = b.mem();
        std::vector<uint32_t> flat(std::max<uint64_t>(rows.back(), 1) * sketch_size_);
        if (b.on_device()) {
            DeviceBuffer d_rows(ctx_, rows.size() * 8), d_sig(ctx_, flat.size() * 4);
            d_rows.upload(rows.data(), rows.size() * 8);
            ctx_.check(kmu_sketch(ctx_.raw(), &p, b.dev_bytes, b.dev_offsets, nullptr, b.n(), d_rows.as<uint64_t>(), d_sig.data(),
                                  nullptr));
            d_sig.download(flat.data(), flat.size() * 4);
        } else {
            ctx_.check(kmu_sketch(ctx_.raw(), &p, b.bytes.data(), b.offsets.data(), b.packed_ptr(), b.n(), rows.data(),
                                  flat.data(), nullptr));
        }
        std::vector<BlockSketchedSeq> out(b.n());
        for (uint32_t i = 0; i < b.n(); i++) {
            out[i].numseq = uint32_t(numfirst + i);
            for (uint64_t r = rows[i]; r < rows[i + 1]; r++)
                out[i].sketch.push_back(BlockSketched{uint32_t(numfirst + i), uint32_t(r - rows[i]),
                                                      {flat.begin() + r * sketch_size_, flat.begin() + (r + 1) * sketch_size_}});
        }
        return out;
    }

    /// create_signature_dump: magic u32, sig_size as ONE byte, sketch_size, kmer_size, block_size u32 = 17 bytes
    /// (seqblocksketch.rs:172-226)
    std::ofstream create_signature_dump(const std::string &dumpfname) const {
        std::ofstream out(dumpfname, std::ios::binary);
        if (!out) throw std::runtime_error("create_signature_dump: cannot open " + dumpfname);
        const uint32_t magic = MAGIC_BLOCKSIG_DUMP, tail[3] = {uint32_t(sketch_size_), uint32_t(kmer_size_), uint32_t(block_size_)};
        const uint8_t sig_size = 4;
        out.write(reinterpret_cast<const char *>(&magic), 4);
        out.write(reinterpret_cast<const char *>(&sig_size), 1);
        out.write(reinterpret_cast<const char *>(tail), sizeof tail);
        return out;
    }
    /// dump_blocks: per sequence numseq u32, nbblock u32, then BlockSketched::dump of every block
    static void dump_blocks(std::ostream &out, const std::vector<BlockSketchedSeq> &seqs) {
        for (const BlockSketchedSeq &s : seqs) {
            const uint32_t nb = uint32_t(s.sketch.size());
            out.write(reinterpret_cast<const char *>(&s.numseq), 4);
            out.write(reinterpret_cast<const char *>(&nb), 4);
            for (const BlockSketched &b : s.sketch) b.dump(out);
        }
    }
    static constexpr uint32_t MAGIC_BLOCKSIG_DUMP = 0xceabbadd;   // seqblocksketch.rs:33

  private:
    size_t block_size_, kmer_size_, sketch_size_;
    Context &ctx_;
};

// =====================================================================================================================
// bottom-k with multiplicities (minhash.rs)
// =====================================================================================================================

struct InvHashCount {   // hashed.rs: (hash, count)
    uint64_t hashed;
    uint16_t count;
};

struct MinHashDist {   // minhash.rs:134-190: containment, jaccard, common, total
    double containment, jaccard;
    uint64_t common, total;
};

/// MinInvHashCountKmer<Kmer, H>: the `size` smallest int64_hash(value) with u16 multiplicities (minhash.rs:194-290)
template <class Kmer> class MinInvHashCountKmer {
  public:
    explicit MinInvHashCountKmer(size_t size, Context &ctx = Context::global()) : size_(size), ctx_(ctx) {}
    /// sketch_kmer_slice: add k-mers (their compressed values)
    void sketch_kmer_slice(const std::vector<Kmer> &kmers) {
        for (const Kmer &k : kmers) pending_.push_back(typename Kmer::Val(k.get_compressed_value()));
    }
    /// get_sketchcount: ascending hashes (minhash.rs:275-289)
    std::vector<InvHashCount> get_sketchcount() const {
        using Val = typename Kmer::Val;
        std::vector<Val> vals = pending_;
        if (vals.empty()) return {};
        const uint64_t off[2] = {0, vals.size()};
        kmu_sketch_params p = detail::sketch_params(KMU_ALGO_BOTTOMK, Kmer::kmu_type, 0, size_, KMU_SIG_U64,
                                                    KMU_HASHER_INT64HASH, KMU_FHASH_IDENTITY_RAW, 0, KMU_MODE_PER_SEQ,
                                                    KMU_INPUT_ASCII);
        std::vector<uint64_t> h(size_);
        std::vector<uint32_t> c(size_);
        ctx_.check(kmu_sketch_hashed(ctx_.raw(), &p, vals.data(), off, 1, h.data(), c.data()));
        std::vector<InvHashCount> out;
        for (size_t i = 0; i < size_ && h[i] != UINT64_MAX; i++) out.push_back({h[i], uint16_t(c[i])});
        return out;
    }
    std::vector<uint64_t> get_hashes_padded() const {
        std::vector<uint64_t> h(size_, UINT64_MAX);
        auto sc = get_sketchcount();
        for (size_t i = 0; i < sc.size(); i++) h[i] = sc[i].hashed;
        return h;
    }
    size_t size() const { return size_; }

  private:
    size_t size_;
    Context &ctx_;
    std::vector<typename Kmer::Val> pending_;
};

/// minhash_distance / mininvhash_distance of two bottom-k sketches (minhash.rs:134-190, 295-340)
template <class Kmer>
MinHashDist mininvhash_distance(const MinInvHashCountKmer<Kmer> &a, const MinInvHashCountKmer<Kmer> &b,
                                Context &ctx = Context::global()) {
    if (a.size() != b.size()) throw std::invalid_argument("sketches of different sizes");
    auto ha = a.get_hashes_padded(), hb = b.get_hashes_padded();
    const uint32_t zero = 0;
    uint32_t out[3] = {0, 0, 0};
    ctx.check(kmu_minhash_distance_pairs(ctx.raw(), ha.data(), 1, hb.data(), 1, uint32_t(a.size()), &zero, &zero, 1,
                                         KMU_MEM_HOST, out));
    return MinHashDist{double(out[0]) / double(out[2]), double(out[0]) / double(out[1]), out[0], out[1]};
}

/// HashCount<u32> of seqminhash.rs: a kept hash and its multiplicity
struct HashCount {
    uint64_t hashed;
    uint16_t count;
};

namespace detail {
inline int range_kmer_type(size_t kmer_size) {
    if (kmer_size == 16) return KMU_KMER16B32BIT;
    if (kmer_size >= 9 && kmer_size <= 15) return KMU_KMER32BIT;
    throw std::invalid_argument("sketch_seqrange: unimplemented kmer_size");   // upstream panics (seqminhash.rs:53, :111)
}
/// the bases [start, end) of a packed sequence as a one-sequence ASCII batch (KmerSeqIterator::set_range keeps the k-mers
/// that lie inside the range)
inline Batch range_batch(const Sequence &seq, size_t start, size_t end) {
    if (end <= start || end > seq.size()) throw std::invalid_argument("bad range");   // set_range -> Err -> panic("bad range")
    Batch b;
    b.bytes.resize(end - start + 16, 0);
    for (size_t i = start; i < end; i++) b.bytes[i - start] = Alphabet2b::decode(seq.get_base(i));
    b.offsets = {0, end - start};
    return b;
}
}  // namespace detail

/// sketch_seqrange_superminhash(seq, range, kmer_size, sketch_size) -> Vec<f64> (seqminhash.rs:19-62):
/// SuperMinHash<f64, u32, NoHashHasher> over int32_hash(canonical k-mer); kmer_size 16 (Kmer16b32bit) or 9..15 (Kmer32bit)
inline std::vector<double> sketch_seqrange_superminhash(const Sequence &seq, std::pair<size_t, size_t> range, size_t kmer_size,
                                                        size_t sketch_size, Context &ctx = Context::global()) {
    kmu_sketch_params p = detail::sketch_params(KMU_ALGO_SUPER, detail::range_kmer_type(kmer_size), kmer_size, sketch_size,
                                                KMU_SIG_F64, KMU_HASHER_NOHASH, KMU_FHASH_CANON_INVHASH, 0, KMU_MODE_PER_SEQ,
                                                KMU_INPUT_ASCII);
    detail::Batch b = detail::range_batch(seq, range.first, range.second);
    std::vector<double> sig(sketch_size);
    ctx.check(kmu_sketch(ctx.raw(), &p, b.bytes.data(), b.offsets.data(), nullptr, 1, nullptr, sig.data(), nullptr));
    return sig;
}

/// sketch_seqrange_minhash(...) -> Vec<HashCount<u32>> (seqminhash.rs:65-119): MinHashCount<u32, NoHashHasher> over the same
/// values; returned ascending by hash (upstream: heap order)
inline std::vector<HashCount> sketch_seqrange_minhash(const Sequence &seq, std::pair<size_t, size_t> range, size_t kmer_size,
                                                      size_t sketch_size, Context &ctx = Context::global()) {
    kmu_sketch_params p = detail::sketch_params(KMU_ALGO_BOTTOMK, detail::range_kmer_type(kmer_size), kmer_size, sketch_size,
                                                KMU_SIG_U64, KMU_HASHER_NOHASH, KMU_FHASH_CANON_INVHASH, 0, KMU_MODE_PER_SEQ,
                                                KMU_INPUT_ASCII);
    detail::Batch b = detail::range_batch(seq, range.first, range.second);
    std::vector<uint64_t> h(sketch_size);
    std::vector<uint32_t> c(sketch_size);
    ctx.check(kmu_sketch(ctx.raw(), &p, b.bytes.data(), b.offsets.data(), nullptr, 1, nullptr, h.data(), c.data()));
    std::vector<HashCount> out;
    for (size_t i = 0; i < sketch_size && h[i] != UINT64_MAX; i++) out.push_back({h[i], uint16_t(c[i])});
    return out;
}

/// minhash_distance(&sk1, &sk2) -> MinHashDist (minhash.rs:134-190)
inline MinHashDist minhash_distance(const std::vector<HashCount> &sk1, const std::vector<HashCount> &sk2,
                                    Context &ctx = Context::global()) {
    const size_t m = std::max(sk1.size(), sk2.size());
    std::vector<uint64_t> ha(std::max<size_t>(m, 1), UINT64_MAX), hb(std::max<size_t>(m, 1), UINT64_MAX);
    for (size_t i = 0; i < sk1.size(); i++) ha[i] = sk1[i].hashed;
    for (size_t i = 0; i < sk2.size(); i++) hb[i] = sk2[i].hashed;
    const uint32_t zero = 0;
    uint32_t out[3] = {0, 0, 0};
    ctx.check(kmu_minhash_distance_pairs(ctx.raw(), ha.data(), 1, hb.data(), 1, uint32_t(ha.size()), &zero, &zero, 1,
                                         KMU_MEM_HOST, out));
    return MinHashDist{double(out[0]) / double(out[2]), double(out[0]) / double(out[1]), out[0], out[1]};
}

// =====================================================================================================================
// counting (kmercount.rs)
// =====================================================================================================================

/// trait KmerCountT (kmercount.rs:48-59)
template <class Kmer> class KmerCountT {
  public:
    virtual ~KmerCountT() = default;
    virtual void insert_kmer(Kmer kmer) = 0;
    virtual uint32_t get_count(Kmer kmer) = 0;
    virtual uint64_t get_nb_distinct() = 0;
    virtual uint64_t get_nb_unique() = 0;
};

/// KmerCounter::new(fpr, capacity, nb_bits) (kmercount.rs:70-123).  The device table is exact: the reference's contract
/// without the false positives of its cuckoo + counting-Bloom pair, so `fpr` is accepted and unused.  Single insertions
/// are queued on the host and reach the device in batches (at the latest when something is read back).
template <class Kmer> class KmerCounter : public KmerCountT<Kmer> {
  public:
    KmerCounter(double /*fpr*/, size_t capacity, uint8_t nb_bits, Context &ctx = Context::global())
        : capacity_(capacity), nb_bits_(nb_bits), ctx_(ctx) {}
    ~KmerCounter() override {
        if (counter_) kmu_count_destroy(counter_);
    }
    KmerCounter(const KmerCounter &) = delete;
    KmerCounter &operator=(const KmerCounter &) = delete;

    void insert_kmer(Kmer kmer) override {
        ensure(kmer.get_nb_base());
        pending_.push_back(uint64_t(kmer.get_compressed_value()));
        if (pending_.size() >= (size_t(1) << 20)) flush();
    }
    /// every canonical k-mer of every read: kmer.reverse_complement().min(kmer) (kmercount.rs:313, 938)
    void insert_reads(const std::vector<const Sequence *> &seqvec, uint8_t kmer_size) {
        insert_reads(detail::gather(seqvec), kmer_size);
    }
    void insert_reads(const detail::Batch &b, uint8_t kmer_size) {
        ensure(kmer_size);
        flush();
        ctx_.check(kmu_count_add_reads(counter_, b.bytes_ptr(), b.offsets_ptr(), b.packed_ptr(), b.n(), b.input_kind, b.mem()));
    }
    uint32_t get_count(Kmer kmer) override {
        ensure(kmer.get_nb_base());
        flush();
        const uint64_t key = kmer.get_compressed_value();
        uint32_t c = 0;
        ctx_.check(kmu_count_query(counter_, &key, 1, KMU_MEM_HOST, &c));
        return c;
    }
    std::vector<uint32_t> get_count(const std::vector<Kmer> &kmers) {
        if (kmers.empty()) return {};
        ensure(kmers[0].get_nb_base());
        flush();
        std::vector<uint64_t> keys;
        for (const Kmer &k : kmers) keys.push_back(k.get_compressed_value());
        std::vector<uint32_t> c(keys.size());
        ctx_.check(kmu_count_query(counter_, keys.data(), keys.size(), KMU_MEM_HOST, c.data()));
        return c;
    }
    uint64_t get_nb_distinct() override {
        if (!counter_) return 0;
        flush();
        uint64_t n = 0;
        ctx_.check(kmu_count_nb_distinct(counter_, &n));
        return n;
    }
    uint64_t get_nb_unique() override {
        if (!counter_) return 0;
        flush();
        uint64_t n = 0;
        ctx_.check(kmu_count_nb_unique(counter_, &n));
        return n;
    }
    /// get_above2_count(kmer) (kmercount.rs:100-105): the count if the k-mer was seen at least twice, else 0
    uint32_t get_above2_count(Kmer kmer) {
        const uint32_t c = get_count(kmer);
        return c >= 2 ? c : 0;
    }
    /// eliminate_once_kmer (kmercount.rs:110-117): forget the k-mers seen once
    void eliminate_once_kmer() {
        if (!counter_) return;
        flush();
        ctx_.check(kmu_count_eliminate_once(counter_));
    }
    uint8_t get_count_nb_bits() const { return nb_bits_; }
    /// (canonical value, count) of the k-mers seen at least twice, sorted by value: what dump_kmer_counter writes
    std::pair<std::vector<uint64_t>, std::vector<uint32_t>> above2_entries() {
        if (!counter_) return {};
        flush();
        uint64_t n = 0;
        ctx_.check(kmu_count_dump(counter_, 2, nullptr, nullptr, 0, &n));
        std::vector<uint64_t> k(std::max<uint64_t>(n, 1));
        std::vector<uint32_t> c(std::max<uint64_t>(n, 1));
        ctx_.check(kmu_count_dump(counter_, 2, k.data(), c.data(), n, &n));
        k.resize(n);
        c.resize(n);
        return {std::move(k), std::move(c)};
    }
    uint8_t kmer_size() const { return kmer_size_; }
    /// the device counter with everything inserted so far (null before the first insertion)
    kmu_counter *raw() {
        if (counter_) flush();
        return counter_;
    }

  private:
    void ensure(uint8_t k) {
        if (counter_) return;
        kmu_count_params p{};
        p.kmer_type = Kmer::kmu_type;
        p.kmer_size = k;
        p.counter_bits = nb_bits_;
        p.capacity_hint = capacity_;
        ctx_.check(kmu_count_create(ctx_.raw(), &p, &counter_));
        kmer_size_ = k;
    }
    void flush() {
        if (pending_.empty()) return;
        ctx_.check(kmu_count_add_kmers(counter_, pending_.data(), pending_.size(), KMU_MEM_HOST));
        pending_.clear();
    }
    size_t capacity_;
    uint8_t nb_bits_;
    Context &ctx_;
    kmu_counter *counter_ = nullptr;
    uint8_t kmer_size_ = 0;
    std::vector<uint64_t> pending_;
};

/// KmerCounterPool (kmercount.rs:424-565): upstream holds one counter per consumer thread and re-dispatches queries by
/// `intNN_hash(kmer) % n`; one GPU holds the whole key space, so the pool is one counter.
template <class Kmer> class KmerCounterPool {
  public:
    KmerCounterPool(size_t capacity, uint8_t nb_bits, Context &ctx = Context::global()) : counter_(0.03, capacity, nb_bits, ctx) {}
    uint32_t get_count(Kmer kmer) { return counter_.get_count(kmer); }
    uint64_t get_nb_distinct() { return counter_.get_nb_distinct(); }
    uint64_t get_nb_unique() { return counter_.get_nb_unique(); }
    uint32_t get_above2_count(Kmer kmer) { return counter_.get_above2_count(kmer); }   // kmercount.rs:439-444
    uint8_t get_count_nb_bits() const { return counter_.get_count_nb_bits(); }
    std::pair<std::vector<uint64_t>, std::vector<uint32_t>> above2_entries() { return counter_.above2_entries(); }
    KmerCounter<Kmer> &counter() { return counter_; }

    /// dump_kmer_counter (kmercount.rs:467-531): COUNTER_MULTIPLE u32, kmer_size u8, bytes per count u8 (= 1), number of
    /// k-mers u64, then per k-mer `Kmer::dump` + count u8.  Records come sorted by value (upstream: order of first
    /// occurrence; the order carries no meaning for a reader).  Returns the number of records.
    size_t dump_kmer_counter(const std::string &fname) {
        auto [kmers, counts] = counter_.above2_entries();
        std::ofstream out(fname, std::ios::binary);
        if (!out) throw std::runtime_error("dump_kmer_counter: cannot open " + fname);
        const uint32_t magic = COUNTER_MULTIPLE;
        const uint8_t k = counter_.kmer_size(), nbc = 1;
        const uint64_t n = kmers.size();
        out.write(reinterpret_cast<const char *>(&magic), 4);
        out.write(reinterpret_cast<const char *>(&k), 1);
        out.write(reinterpret_cast<const char *>(&nbc), 1);
        out.write(reinterpret_cast<const char *>(&n), 8);
        for (size_t i = 0; i < kmers.size(); i++) {
            if constexpr (sizeof(typename Kmer::Val) == 8) {   // Kmer64bit::dump: size byte + u64 (kmer64bit.rs:98-104)
                out.write(reinterpret_cast<const char *>(&k), 1);
                out.write(reinterpret_cast<const char *>(&kmers[i]), 8);
            } else {   // the u32 word `.0` (kmer32bit.rs:141-144)
                const uint32_t w = Kmer::build(uint32_t(kmers[i]), k).raw();
                out.write(reinterpret_cast<const char *>(&w), 4);
            }
            const uint8_t c = uint8_t(std::min<uint32_t>(counts[i], 255));
            out.write(reinterpret_cast<const char *>(&c), 1);
        }
        return kmers.size();
    }
    static constexpr uint32_t COUNTER_MULTIPLE = 0xcea2bbff;   // kmercount.rs:41

  private:
    KmerCounter<Kmer> counter_;
};

/// KmerFilter1 (kmercount.rs:985-1089): which 16-mers occur exactly once in a read set, and where.  Upstream keeps two cuckoo
/// filters (approximate, randomised per process); the exact device table answers the same questions.
class KmerFilter1 {
  public:
    struct OnceKmers {   // records of dump_in_file_once_kmer16b32bit, in file order
        std::vector<uint64_t> kmin;
        std::vector<uint32_t> numseq, numkmer;
    };
    KmerFilter1(uint8_t size, uint32_t capacity, Context &ctx = Context::global()) : counter_(0.03, capacity, 8, ctx), ctx_(ctx) {
        if (size != 16) throw std::invalid_argument("KmerFilter1 holds Kmer16b32bit");
    }
    /// insert_kmer16b32bit(kmer): the caller passes the canonical k-mer (kmercount.rs:1108-1109)
    void insert_kmer16b32bit(Kmer16b32bit kmer) { counter_.insert_kmer(kmer); }
    void insert_reads(const detail::Batch &reads) { counter_.insert_reads(reads, 16); }
    uint64_t get_nb_once() { return counter_.get_nb_unique(); }   // once_f.len()
    OnceKmers once_positions(const detail::Batch &reads) {
        const detail::Batch b = detail::unpacked(reads);   // kmu_count_once_positions takes unpacked bases
        OnceKmers r;
        kmu_counter *c = counter_.raw();
        if (!c) return r;
        uint64_t n = 0;
        ctx_.check(kmu_count_once_positions(c, b.bytes_ptr(), b.offsets_ptr(), b.n(), b.mem(), nullptr, nullptr, nullptr, 0, &n));
        r.kmin.resize(n); r.numseq.resize(n); r.numkmer.resize(n);
        if (n == 0) return r;
        if (b.on_device()) {
            DeviceBuffer dk(ctx_, n * 8), ds(ctx_, n * 4), dp(ctx_, n * 4);
            ctx_.check(kmu_count_once_positions(c, b.dev_bytes, b.dev_offsets, b.n(), KMU_MEM_DEVICE, dk.as<uint64_t>(), ds.as<uint32_t>(),
                                                dp.as<uint32_t>(), n, &n));
            dk.download(r.kmin.data(), n * 8); ds.download(r.numseq.data(), n * 4); dp.download(r.numkmer.data(), n * 4);
        } else {
            ctx_.check(kmu_count_once_positions(c, b.bytes.data(), b.offsets.data(), b.n(), KMU_MEM_HOST, r.kmin.data(),
                                                r.numseq.data(), r.numkmer.data(), n, &n));
        }
        return r;
    }
    /// dump_in_file_once_kmer16b32bit(fname, seqvec) (kmercount.rs:1031-1082): COUNTER_UNIQUE u32, kmer_size u8, number of
    /// k-mers u64, then (kmer u32, numseq u32, numkmer u32) per record.  Returns the number of k-mers dumped.
    size_t dump_in_file_once_kmer16b32bit(const std::string &fname, const detail::Batch &seqvec) {
        OnceKmers r = once_positions(seqvec);
        std::ofstream out(fname, std::ios::binary);
        if (!out) throw std::runtime_error("cannot open " + fname);
        const uint32_t magic = COUNTER_UNIQUE;
        const uint8_t k = 16;
        const uint64_t n = r.kmin.size();
        out.write(reinterpret_cast<const char *>(&magic), 4);
        out.write(reinterpret_cast<const char *>(&k), 1);
        out.write(reinterpret_cast<const char *>(&n), 8);
        for (size_t i = 0; i < r.kmin.size(); i++) {
            const uint32_t rec[3] = {uint32_t(r.kmin[i]), r.numseq[i], r.numkmer[i]};
            out.write(reinterpret_cast<const char *>(rec), 12);
        }
        return r.kmin.size();
    }
    size_t dump_in_file_once_kmer16b32bit(const std::string &fname, const std::vector<Sequence> &seqvec) {
        return dump_in_file_once_kmer16b32bit(fname, detail::gather(detail::pointers(seqvec)));
    }
    static constexpr uint32_t COUNTER_UNIQUE = 0xcea2bbdd;   // kmercount.rs:35

  private:
    KmerCounter<Kmer16b32bit> counter_;
    Context &ctx_;
};

/// filter1_kmer_16b32bit(&seqvec) -> KmerFilter1 (kmercount.rs:1093-1123)
inline std::unique_ptr<KmerFilter1> filter1_kmer_16b32bit(const std::vector<Sequence> &seqvec, Context &ctx = Context::global()) {
    uint64_t bases = 0;
    for (const Sequence &s : seqvec) bases += s.size();
    auto f = std::make_unique<KmerFilter1>(16, uint32_t(std::min<uint64_t>(std::max<uint64_t>(bases, 1024), 0xFFFFFFFFu)), ctx);
    f->insert_reads(detail::gather(detail::pointers(seqvec)));
    return f;
}

/// count_kmer_threaded_one_to_many(seqvec, nb_threads, count_size, kmer_size) -> KmerCounterPool (kmercount.rs:881-974).
/// `nb_threads` is kept for the signature (inside one GPU there is no key-space dispatch); `count_size` = bits per counter.
template <class Kmer>
std::unique_ptr<KmerCounterPool<Kmer>> count_kmer_threaded_one_to_many(const std::vector<Sequence> &seqvec, size_t /*nb_threads*/,
                                                                        size_t count_size, uint8_t kmer_size,
                                                                        Context &ctx = Context::global()) {
    uint64_t bases = 0;
    for (const Sequence &s : seqvec) bases += s.size();
    auto pool = std::make_unique<KmerCounterPool<Kmer>>(std::max<uint64_t>(bases, 1024), uint8_t(count_size), ctx);
    pool->counter().insert_reads(detail::pointers(seqvec), kmer_size);
    return pool;
}

namespace detail {
template <class Kmer> std::unique_ptr<KmerCounter<Kmer>> count_kmer(const std::vector<Sequence> &seqvec, uint8_t kmer_size, Context &ctx) {
    uint64_t bases = 0;
    for (const Sequence &s : seqvec) bases += s.size();
    // upstream sizes its filters for 1.2e9 k-mers (kmercount.rs:303); the exact table only needs the batch
    auto counter = std::make_unique<KmerCounter<Kmer>>(0.03, std::max<uint64_t>(bases, 1024), 8, ctx);
    counter->insert_reads(pointers(seqvec), kmer_size);
    return counter;
}
}  // namespace detail

/// count_kmer16b32bit(&seqvec) (kmercount.rs:332-338): every canonical 16-mer, 8-bit counts
inline std::unique_ptr<KmerCounter<Kmer16b32bit>> count_kmer16b32bit(const std::vector<Sequence> &seqvec, Context &ctx = Context::global()) {
    return detail::count_kmer<Kmer16b32bit>(seqvec, 16, ctx);
}
/// count_kmer32bit(&seqvec, kmer_size) (kmercount.rs:340-347): k <= 14
inline std::unique_ptr<KmerCounter<Kmer32bit>> count_kmer32bit(const std::vector<Sequence> &seqvec, size_t kmer_size,
                                                              Context &ctx = Context::global()) {
    if (kmer_size >= 15) throw std::invalid_argument("count_kmer32bit cannot count kmer of size greater than 14");   // panics upstream
    return detail::count_kmer<Kmer32bit>(seqvec, uint8_t(kmer_size), ctx);
}
/// count_kmer64bit(&seqvec, kmer_size) (kmercount.rs:351-362): 16 < k <= 32 upstream; k = 32 is broken there, refused here
inline std::unique_ptr<KmerCounter<Kmer64bit>> count_kmer64bit(const std::vector<Sequence> &seqvec, size_t kmer_size,
                                                              Context &ctx = Context::global()) {
    if (kmer_size > 32) throw std::invalid_argument("count_kmer64bit cannot count kmer of size greater than 32");
    if (kmer_size <= 16) throw std::invalid_argument("count_kmer64bit is too expensive, use count_kmer32bit");
    return detail::count_kmer<Kmer64bit>(seqvec, uint8_t(kmer_size), ctx);
}
/// count_kmer_thread_independant(seqvec, nb_threads, kmer_size) (kmercount.rs:797-867): upstream's other threaded driver
/// (reads split over threads, k-mers dispatched by key); one GPU holds the whole key space, so the same pool comes out
template <class Kmer>
std::unique_ptr<KmerCounterPool<Kmer>> count_kmer_thread_independant(const std::vector<Sequence> &seqvec, size_t nb_threads,
                                                                      uint8_t kmer_size, Context &ctx = Context::global()) {
    return count_kmer_threaded_one_to_many<Kmer>(seqvec, nb_threads, 8, kmer_size, ctx);
}

/// KmerCountReload (kmercount.rs:1132-1500): a dump of the counter read back -- for up-to-32-bit k-mers, as upstream
struct KmerCoord {   // kmercount.rs: read_num, pos
    uint32_t read_num, pos;
};
enum class CounterType { Unique, Multiple };
class KmerCountReload {
  public:
    uint32_t get_kmer_size() const { return kmer_size_; }
    size_t get_nb_kmer() const { return nb_kmer_; }
    /// get_coord_from_rank (kmercount.rs:1479-1486)
    std::optional<KmerCoord> get_coord_from_rank(size_t rank) const {
        if (rank < positions_.size()) return positions_[rank];
        return std::nullopt;
    }
    /// get_multi_kmer_counts (kmercount.rs:1489-1500): here in file order
    std::optional<std::vector<uint16_t>> get_multi_kmer_counts() const {
        if (type_ != CounterType::Multiple) return std::nullopt;
        return counts_;
    }
    const std::vector<uint32_t> &kmers() const { return kmers_; }   // the u32 words as dumped (Kmer32bit carries k in its top nibble)
    /// load_multiple_kmers_from_file (kmercount.rs:1209-1351): COUNTER_MULTIPLE, kmer_size u8, bytes per count u8, number u64,
    /// then (kmer u32, count) until the end of the file (upstream's header count is approximate, :1200-1207)
    static std::unique_ptr<KmerCountReload> load_multiple_kmers_from_file(const std::string &fname) {
        std::ifstream in(fname, std::ios::binary);
        uint32_t magic = 0;
        uint8_t k = 0, nbc = 0;
        uint64_t n = 0;
        in.read(reinterpret_cast<char *>(&magic), 4); in.read(reinterpret_cast<char *>(&k), 1);
        in.read(reinterpret_cast<char *>(&nbc), 1); in.read(reinterpret_cast<char *>(&n), 8);
        if (!in || magic != 0xcea2bbff || (nbc != 1 && nbc != 2) || k > 16) return nullptr;
        auto r = std::unique_ptr<KmerCountReload>(new KmerCountReload(CounterType::Multiple, k, n));
        for (;;) {
            uint32_t kmer = 0;
            uint16_t c = 0;
            in.read(reinterpret_cast<char *>(&kmer), 4);
            in.read(reinterpret_cast<char *>(&c), nbc);
            if (!in) break;
            r->kmers_.push_back(kmer);
            r->counts_.push_back(c);
        }
        return r;
    }
    /// load_unique_kmer_from_file (kmercount.rs:1356-1470): COUNTER_UNIQUE, kmer_size u8, number u64, then (kmer u32, numseq u32,
    /// numkmer u32)
    static std::unique_ptr<KmerCountReload> load_unique_kmer_from_file(const std::string &fname) {
        std::ifstream in(fname, std::ios::binary);
        uint32_t magic = 0;
        uint8_t k = 0;
        uint64_t n = 0;
        in.read(reinterpret_cast<char *>(&magic), 4); in.read(reinterpret_cast<char *>(&k), 1); in.read(reinterpret_cast<char *>(&n), 8);
        if (!in || magic != 0xcea2bbdd) return nullptr;
        auto r = std::unique_ptr<KmerCountReload>(new KmerCountReload(CounterType::Unique, k, n));
        for (;;) {
            uint32_t rec[3];
            in.read(reinterpret_cast<char *>(rec), 12);
            if (!in) break;
            r->kmers_.push_back(rec[0]);
            r->positions_.push_back(KmerCoord{rec[1], rec[2]});
        }
        return r;
    }

  private:
    KmerCountReload(CounterType t, uint32_t k, size_t n) : type_(t), kmer_size_(k), nb_kmer_(n) {}
    CounterType type_;
    uint32_t kmer_size_;
    size_t nb_kmer_;
    std::vector<uint32_t> kmers_;
    std::vector<KmerCoord> positions_;
    std::vector<uint16_t> counts_;
};

// =====================================================================================================================
// io: the reader rule of datasketcher / parsefastq, on the device
// =====================================================================================================================

/// Accepted reads of a FASTQ text as one byte array + offsets (the form every kernel consumes), plus the reader's tallies
/// (io.rs:37-57, datasketcher.rs:358-388: a record with a byte outside ACGTacgt is dropped and counted).
struct FastqReads {
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets;
    kmu_ingest_info info{};
    size_t nb_reads() const { return offsets.empty() ? 0 : offsets.size() - 1; }
};

/// reads [first, last) as a batch for the sketchers / the counter
inline detail::Batch batch_of(const FastqReads &r, size_t first, size_t last) {
    detail::Batch b;
    b.offsets.resize(last - first + 1);
    for (size_t i = first; i <= last; i++) b.offsets[i - first] = r.offsets[i] - r.offsets[first];
    b.bytes.assign(r.bases.begin() + r.offsets[first], r.bases.begin() + r.offsets[last]);
    b.bytes.resize(b.bytes.size() + 16, 0);
    return b;
}

/// FASTQ or FASTA by the first byte, like needletail::parse_fastx_file (io.rs:37)
inline FastqReads parse_fastx_text(const uint8_t *text, size_t n, Context &ctx = Context::global()) {
    FastqReads r;
    ctx.check(kmu_ingest_fastx(ctx.raw(), text, n, KMU_MEM_HOST, nullptr, 0, nullptr, 0, nullptr, &r.info));
    r.bases.resize(r.info.kept_bases + 16);
    r.offsets.resize(r.info.n_kept + 1);
    ctx.check(kmu_ingest_fastx(ctx.raw(), text, n, KMU_MEM_HOST, r.bases.data(), r.bases.size(), r.offsets.data(),
                               r.offsets.size(), nullptr, &r.info));
    return r;
}
inline FastqReads parse_fastq_text(const uint8_t *text, size_t n, Context &ctx = Context::global()) {
    return parse_fastx_text(text, n, ctx);
}

/// The same, with the accepted reads LEFT ON THE DEVICE: the file's text goes up once, the reads never come back --
/// sketchers and counters take ranges of them (`batch(first, last)`) and only signatures / counts cross PCIe.
class DeviceReads {
  public:
    kmu_ingest_info info{};
    std::vector<uint64_t> offsets;   // host copy (n + 1): sizes, pack boundaries, block layouts
    size_t nb_reads() const { return offsets.empty() ? 0 : offsets.size() - 1; }
    /// reads [first, last): no copy, the device offsets index the whole base array
    detail::Batch batch(size_t first, size_t last) const {
        detail::Batch b;
        b.offsets.assign(offsets.begin() + first, offsets.begin() + last + 1);
        b.dev_bytes = bases_.as<uint8_t>();
        b.dev_offsets = offsets_dev_.as<uint64_t>() + first;
        return b;
    }
    /// FASTQ / FASTA text already on the device (16-byte aligned) -> accepted reads on the device
    static DeviceReads from_device_text(Context &ctx, const uint8_t *d_text, uint64_t n) {
        DeviceReads r;
        ctx.check(kmu_ingest_fastx(ctx.raw(), d_text, n, KMU_MEM_DEVICE, nullptr, 0, nullptr, 0, nullptr, &r.info));
        r.bases_ = DeviceBuffer(ctx, r.info.kept_bases + 64);
        r.offsets_dev_ = DeviceBuffer(ctx, (r.info.n_kept + 1) * 8);
        ctx.check(kmu_ingest_fastx(ctx.raw(), d_text, n, KMU_MEM_DEVICE, r.bases_.as<uint8_t>(), r.bases_.bytes(),
                                   r.offsets_dev_.as<uint64_t>(), r.info.n_kept + 1, nullptr, &r.info));
        r.offsets.resize(r.info.n_kept + 1);
        r.offsets_dev_.download(r.offsets.data(), r.offsets.size() * 8);
        return r;
    }
    /// The file goes up in slabs: `n_readers` threads read slabs of it in parallel (pread), each slab is uploaded as soon
    /// as it is in memory (uploads are serialised: one context, one thread at a time), so reading and PCIe overlap.
    static DeviceReads from_file(const std::string &fname, Context &ctx = Context::global(), size_t slab = size_t(64) << 20,
                                 unsigned n_readers = 4) {
        const int fd = ::open(fname.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + fname);
        struct stat st;
        if (::fstat(fd, &st) != 0) {
            ::close(fd);
            throw std::runtime_error("cannot stat " + fname);
        }
        const uint64_t n = uint64_t(st.st_size);
        DeviceBuffer text(ctx, n + 64);
        const uint64_t n_slabs = (n + slab - 1) / slab;
        std::atomic<uint64_t> next{0};
        std::mutex upload_lock;
        std::string error;
        auto reader = [&]() {
            std::vector<char> host(std::min<uint64_t>(slab, std::max<uint64_t>(n, 1)));
            for (;;) {
                const uint64_t i = next.fetch_add(1);
                if (i >= n_slabs) return;
                const uint64_t at = i * slab, want = std::min<uint64_t>(slab, n - at);
                uint64_t got = 0;
                while (got < want) {
                    const ssize_t r = ::pread(fd, host.data() + got, want - got, off_t(at + got));
                    if (r <= 0) break;
                    got += uint64_t(r);
                }
                std::lock_guard<std::mutex> g(upload_lock);
                if (got != want) {
                    error = "short read on " + fname;
                    next = n_slabs;
                    return;
                }
                try {
                    text.upload(host.data(), want, at);
                } catch (const std::exception &e) {
                    error = e.what();
                    next = n_slabs;
                    return;
                }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < std::max(1u, n_readers) && t < n_slabs; t++) pool.emplace_back(reader);
        reader();
        for (auto &t : pool) t.join();
        ::close(fd);
        if (!error.empty()) throw std::runtime_error(error);
        return from_device_text(ctx, text.as<uint8_t>(), n);
    }

  private:
    DeviceBuffer bases_, offsets_dev_;
};

inline FastqReads parse_fastq_file(const std::string &fname, Context &ctx = Context::global()) {
    std::ifstream in(fname, std::ios::binary | std::ios::ate);
    if (!in) throw std::runtime_error("cannot open " + fname);
    const std::streamsize n = in.tellg();
    in.seekg(0);
    std::vector<uint8_t> text(size_t(n) + 16);
    in.read(reinterpret_cast<char *>(text.data()), n);
    return parse_fastq_text(text.data(), size_t(n), ctx);
}

}  // namespace kmerutils

#endif  // KMERUTILS_HPP
