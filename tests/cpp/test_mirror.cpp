// Tests of the C++ host mirror (include/kmerutils.hpp) on the GPU.  Each case follows one test of the reference -- same
// strings, same parameters, same assertions (file:line quoted) -- and then checks the device result bit for bit against
// the CPU oracle (oracle/kmu_oracle.h; test infrastructure only).
//
//   ./test_mirror            run everything, print "ok <name>" / "FAIL <name>: why", exit code = number of failures
//   ./test_mirror <name>...  run the named cases
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <map>
#include <random>
#include <sstream>
#include <string>

#include "../../include/kmerutils.hpp"
#include "../../oracle/kmu_oracle.h"

using namespace kmerutils;

namespace {

struct Failure : std::runtime_error {
    using std::runtime_error::runtime_error;
};
#define CHECK(cond)                                                                                                   \
    do {                                                                                                              \
        if (!(cond)) {                                                                                                \
            std::ostringstream os_;                                                                                   \
            os_ << __FILE__ << ":" << __LINE__ << ": " #cond;                                                         \
            throw Failure(os_.str());                                                                                 \
        }                                                                                                             \
    } while (0)

std::vector<std::pair<std::string, std::function<void()>>> &registry() {
    static std::vector<std::pair<std::string, std::function<void()>>> r;
    return r;
}
struct Reg {
    Reg(const char *name, std::function<void()> f) { registry().emplace_back(name, std::move(f)); }
};
#define TEST(name)                                                                                                    \
    void name();                                                                                                      \
    Reg reg_##name(#name, name);                                                                                      \
    void name()

// the 80 bases every DNA test of the reference uses
const std::string SEQSTR = "TCAAAGGGAAACATTCAAAATCAGTATGCGCCCGTTCAGTTACGTATTGCTCTCGCTAATGAGATGGGCTGGGTACAGAG";
const std::string AA1 = "MTEQIELIKLYSTRILALAAQMPHVGSLDNPDASAMKRSPLCGSKVTVDVIMQNGKITFDGFEVLAPASEYKNRHASILLSLDATAEACASIAAQNSA";
const std::string AA2 = "MTEQIELIKLYSTRILALAAQMPHVGSLDNPDASAMKRSPLCGSKVMTEQIELIKLYSTRILALAAQMPHVGSLDNPDASAMKRSPLCGSKV";

// ---- oracle helpers -----------------------------------------------------------------------------------------------------
struct Ascii {
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> off{0};
    void add(const std::vector<uint8_t> &s) {
        bytes.insert(bytes.end(), s.begin(), s.end());
        off.push_back(bytes.size());
    }
    void add(const std::string &s) { add(std::vector<uint8_t>(s.begin(), s.end())); }
};

template <class Sig>
std::vector<std::vector<Sig>> oracle_sketch(const Ascii &a, int algo, int kmer_type, int k, int m, int sig_type, int hasher,
                                            int fhash, int mode = KMU_MODE_PER_SEQ) {
    kmu_sketch_params p{};
    p.algo = algo; p.kmer_type = kmer_type; p.kmer_size = k; p.sketch_size = m; p.sig_type = sig_type; p.hasher = hasher;
    p.fhash = fhash; p.mode = mode; p.input_kind = KMU_INPUT_ASCII;
    const size_t rows = mode == KMU_MODE_ALL_SEQS ? 1 : a.off.size() - 1;
    std::vector<Sig> flat(rows * m);
    std::vector<uint8_t> bytes = a.bytes;
    bytes.resize(bytes.size() + 16);
    int rc = kmo_sketch(&p, bytes.data(), a.off.data(), nullptr, uint32_t(a.off.size() - 1), nullptr, flat.data(), nullptr);
    if (rc) throw Failure("oracle kmo_sketch failed: " + std::to_string(rc));
    std::vector<std::vector<Sig>> out(rows);
    for (size_t r = 0; r < rows; r++) out[r].assign(flat.begin() + r * m, flat.begin() + (r + 1) * m);
    return out;
}

template <class T> double equal_fraction(const std::vector<T> &a, const std::vector<T> &b) {
    size_t inter = 0;
    for (size_t i = 0; i < a.size(); i++) inter += a[i] == b[i];
    return double(inter) / double(a.size());
}

// =========================================================================================================================
// base
// =========================================================================================================================

// sequence.rs: packing order, get_base, reverse complement, decompress; and the device packer agrees
TEST(test_sequence_pack_and_revcomp) {
    Sequence seq(SEQSTR);
    CHECK(seq.size() == 80 && seq.compressed_length() == 20);
    CHECK(seq.packed()[0] == 0b11010000);   // T C A A, first base in bits 7..6 (sequence.rs:48-73)
    auto back = seq.decompress();
    CHECK(std::string(back.begin(), back.end()) == SEQSTR);
    Sequence rc = seq.get_reverse_complement();
    auto rcs = rc.decompress();
    for (size_t i = 0; i < 80; i++) CHECK(rcs[i] == "TGCA"[std::string("ACGT").find(SEQSTR[79 - i])]);
    // device: kmu_pack2b == Sequence::new(raw, 2), also for a length that is no multiple of 4
    for (size_t len : {80u, 77u, 3u}) {
        Sequence s(std::string_view(SEQSTR).substr(0, len));
        const uint64_t off[2] = {0, len};
        std::vector<uint8_t> packed(len / 4 + 32), raw(SEQSTR.begin(), SEQSTR.begin() + len);
        raw.resize(len + 16);
        uint64_t poff[2];
        Context::global().check(kmu_pack2b(Context::global().raw(), raw.data(), off, 1, KMU_MEM_HOST, packed.data(), poff));
        CHECK(poff[1] == s.compressed_length());
        CHECK(std::equal(s.packed().begin(), s.packed().end(), packed.begin()));
    }
    bool threw = false;
    try {
        Sequence bad(std::string_view("ACGNT"));
    } catch (const std::invalid_argument &) {
        threw = true;   // Alphabet2b::encode panics upstream (alphabet.rs:125)
    }
    CHECK(threw);
}

// kmergenerator.rs: every k-mer the device generates equals the chain build / push on the host value type, for the three
// DNA types; reverse complement twice is the identity; canonical = min
template <class Kmer> void check_generator(uint8_t k) {
    Sequence seq(SEQSTR);
    std::vector<Kmer> vkmer = KmerGenerator<Kmer>(k).generate_kmer(seq);
    CHECK(vkmer.size() == 80u - k + 1);
    typename Kmer::Val first = 0;
    for (int i = 0; i < k; i++) first = typename Kmer::Val(first << 2) | seq.get_base(i);
    Kmer cur = Kmer::build(first, k);
    for (size_t p = 0; p < vkmer.size(); p++) {
        if (p) cur = cur.push(seq.get_base(p + k - 1));
        CHECK(cur == vkmer[p]);
        CHECK(vkmer[p].get_nb_base() == k);
        CHECK(vkmer[p].reverse_complement().reverse_complement() == vkmer[p]);
        auto un = vkmer[p].get_uncompressed_kmer();
        CHECK(std::string(un.begin(), un.end()) == SEQSTR.substr(p, k));
    }
    // the k-mers of the reverse complement strand are the reverse complements, in reverse order
    std::vector<Kmer> rck = KmerGenerator<Kmer>(k).generate_kmer(seq.get_reverse_complement());
    for (size_t p = 0; p < vkmer.size(); p++) CHECK(rck[vkmer.size() - 1 - p] == vkmer[p].reverse_complement());
}
TEST(test_kmer_generator_32bit) { check_generator<Kmer32bit>(5); check_generator<Kmer32bit>(14); }
TEST(test_kmer_generator_16b32bit) { check_generator<Kmer16b32bit>(16); }
TEST(test_kmer_generator_64bit) { check_generator<Kmer64bit>(16); check_generator<Kmer64bit>(31); }

// kmergenerator.rs:661-700 (k-mers of a base range), :777-850 (the 31 distinct 3-mers of a 48-base string with their
// multiplicities), :853-894 (weighted 15-mers of a string whose head is repeated) -- through the device
TEST(test_generate_kmer_in_range_and_weighted) {
    const std::string s50 = "TCAAAGGGAAACATTCAAAATCAGTATGCGCCCGTTCAGTTACGTATTGC";
    Sequence seq(s50);
    auto vk = KmerGenerator<Kmer16b32bit>(16).generate_kmer_in_range(seq, 3, 25);
    CHECK(vk.size() == 7);   // :690
    for (size_t i = 0; i < vk.size(); i++) {
        auto un = vk[i].get_uncompressed_kmer();
        CHECK(std::string(un.begin(), un.end()) == s50.substr(3 + i, 16));
    }
    for (auto [b, e] : {std::pair<size_t, size_t>{5, 5}, {9, 3}, {0, 51}}) {   // set_range Err (sequence.rs:563-565)
        bool threw = false;
        try { (void) KmerGenerator<Kmer32bit>(8).generate_kmer_in_range(seq, b, e); } catch (const KmuError &x) { threw = x.status() == KMU_E_BAD_ARG; }
        CHECK(threw);
    }
    const std::string s48 = s50.substr(0, 48);
    auto w3 = KmerGenerator<Kmer32bit>(3).generate_weighted_kmer(Sequence(s48));
    CHECK(w3.size() == 31);   // :822
    std::map<std::string, uint32_t> got;
    for (auto &[km, c] : w3) { auto un = km.get_uncompressed_kmer(); got[std::string(un.begin(), un.end())] = c; }
    CHECK(got["TCA"] == 4 && got["AAA"] == 4 && got["CAA"] == 2 && got["ATT"] == 2 && got["ACG"] == 1 && got["GGG"] == 1);
    uint32_t total = 0;
    for (auto &kv : got) total += kv.second;
    CHECK(total == 46);
    const std::string s72 = "TCAAAGGGAAACATTCAAAATCAGTATGCGCCCGTTCAGTTACGTATTTCAAAGGGAAACATTCAAAATCAG";
    auto w15 = KmerGenerator<Kmer64bit>(15).generate_weighted_kmer(Sequence(s72));
    for (auto &[km, c] : w15) {
        auto un = km.get_uncompressed_kmer();
        const std::string sub(un.begin(), un.end());
        size_t occ = 0;
        for (size_t p = 0; p + 15 <= s72.size(); p++) occ += s72.compare(p, 15, sub) == 0;
        CHECK(occ == c && (c == 1 || c == 2));
    }
}

// kmer16b32bit.rs:145-157 style known answers for the reverse complement
TEST(test_reverse_complement_known_answers) {
    // 16 x A <-> 16 x T
    CHECK(Kmer16b32bit(0).reverse_complement().v == 0xFFFFFFFFu);
    // ACGT is its own reverse complement
    Kmer32bit acgt = Kmer32bit::build(0b00011011, 4);
    CHECK(acgt.reverse_complement() == acgt);
    // AAC -> GTT
    CHECK(Kmer32bit::build(0b000001, 3).reverse_complement() == Kmer32bit::build(0b101111, 3));
    CHECK(Kmer64bit(0b000001, 3).reverse_complement() == Kmer64bit(0b101111, 3));
    // Ord compares the number of bases first (kmer32bit.rs:47-55)
    CHECK(Kmer32bit::build(0xFF, 4) < Kmer32bit::build(0x01, 5));
}

// =========================================================================================================================
// sketching, DNA (seqsketchjaccard.rs tests)
// =========================================================================================================================

// seqsketchjaccard.rs:742-851
TEST(test_pminhasha_kmer_smallb) {
    const size_t kmer_size = 5, sketch_size = 4000;
    Sequence seqa(SEQSTR);
    std::vector<Sequence> vecseqb;
    vecseqb.emplace_back(std::string_view(SEQSTR).substr(0, 40));   // half the length of seqa
    Sequence seqarevcomp = seqa.get_reverse_complement();
    vecseqb.push_back(seqarevcomp);
    const double jac_theo_0 = double(40 - kmer_size) / double(80 - kmer_size);
    auto vecsig = jaccard_index_probminhash3a<Kmer32bit>(seqa, vecseqb, sketch_size, kmer_size, kmer_revcomp_hash_fn);
    CHECK(vecsig[0] >= 0.75 * jac_theo_0);   // :784
    CHECK(vecsig[1] >= 1.);                  // :785
    vecsig = jaccard_index_probminhash3a<Kmer32bit>(seqa, vecseqb, sketch_size, kmer_size, kmer_identity);
    CHECK(vecsig[0] >= 0.75 * jac_theo_0);   // :791
    if (vecsig[1] > 0.) {
        // the k-mers common to seqa and its reverse complement: with k = 5, ACGTA and TACGT (:845)
        SeqSketcher sk(kmer_size, sketch_size);
        auto sigs = sk.sketch_probminhash3a<Kmer32bit>(std::vector<const Sequence *>{&seqa, &seqarevcomp}, kmer_identity);
        auto [jac, common] = probminhash_get_jaccard_objects(sigs[0], sigs[1]);
        CHECK(jac > 0. && common.has_value());
        for (uint32_t v : *common) {
            auto s = Kmer32bit(v).get_uncompressed_kmer();
            std::string str(s.begin(), s.end());
            CHECK(str == "ACGTA" || str == "TACGT");
        }
    }
    CHECK(vecsig[1] <= 0.1);   // :850
    // parity: the four signatures equal the oracle's bit for bit
    Ascii a;
    a.add(SEQSTR); a.add(SEQSTR.substr(0, 40)); a.add(seqarevcomp.decompress());
    SeqSketcher sk(kmer_size, sketch_size);
    std::vector<const Sequence *> all{&seqa, &vecseqb[0], &vecseqb[1]};
    for (FHash f : {kmer_revcomp_hash_fn, kmer_identity})
        CHECK(sk.sketch_probminhash3a<Kmer32bit>(all, f) ==
              oracle_sketch<uint32_t>(a, KMU_ALGO_PROB3A, KMU_KMER32BIT, 5, 4000, KMU_SIG_U32, KMU_HASHER_NOHASH, int(f)));
}

// seqsketchjaccard.rs:854-910
TEST(test_pminhasha_k16b32bit_serial) {
    const size_t kmer_size = 16;
    Sequence seqa(SEQSTR);
    std::vector<Sequence> vecseqb;
    vecseqb.emplace_back(std::string_view(SEQSTR).substr(0, 40));
    vecseqb.push_back(seqa.get_reverse_complement());
    const double jac_theo_0 = double(40 - kmer_size) / double(80 - kmer_size);
    auto vec_0 = jaccard_index_probminhash3a<Kmer16b32bit>(seqa, {vecseqb[0]}, 50, 16, kmer_revcomp_hash_fn);
    auto vec_1 = jaccard_index_probminhash3a<Kmer16b32bit>(seqa, {vecseqb[1]}, 50, 16, kmer_revcomp_hash_fn);
    CHECK(vec_0[0] >= 0.75 * jac_theo_0);   // :902
    CHECK(vec_1[0] >= 1.);                  // :903
    auto vecsig = jaccard_index_probminhash3a<Kmer16b32bit>(seqa, vecseqb, 50, 16, kmer_identity);
    CHECK(vecsig[0] >= 0.75 * jac_theo_0);  // :908
    CHECK(vecsig[1] <= 0.1);                // :909
    Ascii a;
    a.add(SEQSTR); a.add(SEQSTR.substr(0, 40)); a.add(vecseqb[1].decompress());
    std::vector<const Sequence *> all{&seqa, &vecseqb[0], &vecseqb[1]};
    CHECK(SeqSketcher(16, 50).sketch_probminhash3a<Kmer16b32bit>(all, kmer_revcomp_hash_fn) ==
          oracle_sketch<uint32_t>(a, KMU_ALGO_PROB3A, KMU_KMER16B32BIT, 16, 50, KMU_SIG_U32, KMU_HASHER_NOHASH,
                                  KMU_FHASH_CANON_INVHASH));
    // sketch_probminhash3 (:272-319) keeps the same signature
    CHECK(SeqSketcher(16, 50).sketch_probminhash3<Kmer16b32bit>(all, kmer_revcomp_hash_fn) ==
          SeqSketcher(16, 50).sketch_probminhash3a<Kmer16b32bit>(all, kmer_revcomp_hash_fn));
}

// seqsketchjaccard.rs:914-944
TEST(test_pminhash_kmer64bit_serial) {
    const size_t kmer_size = 16;
    Sequence seqa(SEQSTR);
    std::vector<Sequence> vecseqb;
    vecseqb.emplace_back(std::string_view(SEQSTR).substr(0, 40));
    vecseqb.push_back(seqa.get_reverse_complement());
    auto vec_jac = jaccard_index_probminhash3a<Kmer64bit>(seqa, vecseqb, 50, 16, kmer_revcomp_hash_fn);
    const double jac_theo_0 = double(40 - kmer_size) / double(80 - kmer_size);
    CHECK(vec_jac[0] >= 0.75 * jac_theo_0);   // :942
    CHECK(vec_jac[1] >= 1.);                  // :943
    Ascii a;
    a.add(SEQSTR); a.add(SEQSTR.substr(0, 40)); a.add(vecseqb[1].decompress());
    std::vector<const Sequence *> all{&seqa, &vecseqb[0], &vecseqb[1]};
    CHECK(SeqSketcher(16, 50).sketch_probminhash3a<Kmer64bit>(all, kmer_revcomp_hash_fn) ==
          oracle_sketch<uint64_t>(a, KMU_ALGO_PROB3A, KMU_KMER64BIT, 16, 50, KMU_SIG_U64, KMU_HASHER_NOHASH,
                                  KMU_FHASH_CANON_INVHASH));
}

// seqsketchjaccard.rs:947-1010
TEST(test_superminhash_kmer_16b32bit_serial) {
    const size_t kmer_size = 16, sketch_size = 100;
    Sequence seqa(SEQSTR), seqb1(std::string_view(SEQSTR).substr(0, 40));
    Sequence seqarevcomp = seqa.get_reverse_complement();
    std::vector<const Sequence *> vecseq{&seqa, &seqb1, &seqarevcomp};
    const double jac_theo_0 = double(40 - kmer_size) / double(80 - kmer_size);
    SeqSketcher sketcher(kmer_size, sketch_size);
    auto sig_vec = sketcher.sketch_superminhash<Kmer16b32bit>(vecseq, kmer_revcomp_hash_fn);
    double d_01 = compute_superminhash_jaccard(sig_vec[0], sig_vec[1]);
    double d_02 = compute_superminhash_jaccard(sig_vec[0], sig_vec[2]);
    CHECK(d_01 >= 0.75 * jac_theo_0);   // :992
    CHECK(d_02 >= 1.);                  // :993
    Ascii a;
    a.add(SEQSTR); a.add(SEQSTR.substr(0, 40)); a.add(seqarevcomp.decompress());
    CHECK(sig_vec == oracle_sketch<double>(a, KMU_ALGO_SUPER, KMU_KMER16B32BIT, 16, 100, KMU_SIG_F64, KMU_HASHER_FNV1A,
                                           KMU_FHASH_CANON_INVHASH));
    sig_vec = sketcher.sketch_superminhash<Kmer16b32bit>(vecseq, kmer_identity);
    d_01 = compute_superminhash_jaccard(sig_vec[0], sig_vec[1]);
    d_02 = compute_superminhash_jaccard(sig_vec[0], sig_vec[2]);
    CHECK(d_01 >= 0.75 * jac_theo_0);   // :1003
    CHECK(d_02 <= 0.1);                 // :1004
    // f32 signatures
    auto sig32 = sketcher.sketch_superminhash<Kmer16b32bit, float>(vecseq, kmer_revcomp_hash_fn);
    CHECK(sig32 == oracle_sketch<float>(a, KMU_ALGO_SUPER, KMU_KMER16B32BIT, 16, 100, KMU_SIG_F32, KMU_HASHER_FNV1A,
                                        KMU_FHASH_CANON_INVHASH));
}

// trait path: ProbHash3aSketch / SuperHashSketch / SuperHash2Sketch (setsketchert.rs:85-336, 904-1046), per sequence and
// for the whole list, against the oracle
TEST(test_seqsketcher_trait_dna) {
    Sequence seqa(SEQSTR), seqb(std::string_view(SEQSTR).substr(20, 60));
    Sequence seqc = seqa.get_reverse_complement();
    std::vector<const Sequence *> vseq{&seqa, &seqb, &seqc};
    Ascii a;
    a.add(SEQSTR); a.add(SEQSTR.substr(20, 60)); a.add(seqc.decompress());
    SeqSketcherParams args(12, 64, SketchAlgo::PROB3A, DataType::DNA);
    ProbHash3aSketch<Kmer32bit> prob(args);
    CHECK(prob.get_kmer_size() == 12 && prob.get_sketch_size() == 64 && prob.get_algo() == SketchAlgo::PROB3A);
    const SeqSketcherT<Kmer32bit, uint32_t> &as_trait = prob;   // usable through the trait
    CHECK(as_trait.sketch_compressedkmer(vseq, kmer_revcomp_hash_fn) ==
          oracle_sketch<uint32_t>(a, KMU_ALGO_PROB3A, KMU_KMER32BIT, 12, 64, KMU_SIG_U32, KMU_HASHER_NOHASH,
                                  KMU_FHASH_CANON_INVHASH));
    auto all = as_trait.sketch_compressedkmer_seqs(vseq, kmer_revcomp_hash_fn);
    CHECK(all.size() == 1);   // outer length 1 (setsketchert.rs:74-79)
    CHECK(all == oracle_sketch<uint32_t>(a, KMU_ALGO_PROB3A, KMU_KMER32BIT, 12, 64, KMU_SIG_U32, KMU_HASHER_NOHASH,
                                         KMU_FHASH_CANON_INVHASH, KMU_MODE_ALL_SEQS));
    SuperHashSketch<Kmer32bit, double> sup(args);
    CHECK(sup.sketch_compressedkmer(vseq, kmer_revcomp_hash_fn) ==
          oracle_sketch<double>(a, KMU_ALGO_SUPER, KMU_KMER32BIT, 12, 64, KMU_SIG_F64, KMU_HASHER_NOHASH,
                                KMU_FHASH_CANON_INVHASH));
    CHECK(sup.sketch_compressedkmer_seqs(vseq, kmer_revcomp_hash_fn) ==
          oracle_sketch<double>(a, KMU_ALGO_SUPER, KMU_KMER32BIT, 12, 64, KMU_SIG_F64, KMU_HASHER_NOHASH,
                                KMU_FHASH_CANON_INVHASH, KMU_MODE_ALL_SEQS));
    SuperHash2Sketch<Kmer32bit, uint64_t, FnvHasher> sup2(args);
    CHECK(sup2.sketch_compressedkmer(vseq, kmer_revcomp_hash_fn) ==
          oracle_sketch<uint64_t>(a, KMU_ALGO_SUPER2, KMU_KMER32BIT, 12, 64, KMU_SIG_U64, KMU_HASHER_FNV1A,
                                  KMU_FHASH_CANON_INVHASH));
}

// setsketchert.rs:1075-1146 and :1150-1230: sequences of 60 bases sketched to 800 (OptDens) and 8000 (RevOptDens) bins --
// almost every bin is filled by densification, and the estimate still has to come out at the true 0.5
template <template <class, class> class Sketcher, class S> void check_densified(size_t sketch_size, int algo, int sig_type) {
    const std::string str1 = "ATCATGCCCCTTTAGAAAATTTCCGGATCATCGTACGGAGCATGCGTACAACGTCGATGC";
    const std::string str2 = "ATCATGCCCCTTTAGAAAATTTCCGGATCATCATGCCCCTTTAGAAAATTTCCGGATC";
    Sequence seq1(str1), seq2(str2);
    std::vector<const Sequence *> vseq{&seq1, &seq2};
    SeqSketcherParams sketch_args(5, sketch_size, SketchAlgo::OPTDENS, DataType::DNA);
    Sketcher<Kmer32bit, S> sketcher(sketch_args);
    auto signatures = sketcher.sketch_compressedkmer(vseq, kmer_hash_fn);
    CHECK(std::fabs(equal_fraction(signatures[0], signatures[1]) - 0.5) < 1. / 10.);   // :1124 / :1145 / :1208 / :1229
    Ascii a;
    a.add(str1); a.add(str2);
    CHECK(signatures == oracle_sketch<S>(a, algo, KMU_KMER32BIT, 5, int(sketch_size), sig_type, KMU_HASHER_NOHASH,
                                         KMU_FHASH_VALUE_MASKED));
    auto all = sketcher.sketch_compressedkmer_seqs(vseq, kmer_hash_fn);
    CHECK(all.size() == 1);
    CHECK(all == oracle_sketch<S>(a, algo, KMU_KMER32BIT, 5, int(sketch_size), sig_type, KMU_HASHER_NOHASH,
                                  KMU_FHASH_VALUE_MASKED, KMU_MODE_ALL_SEQS));
}
TEST(test_seq_optdensminhash_trait) {
    check_densified<OptDensHashSketch, double>(800, KMU_ALGO_OPTDENS, KMU_SIG_F64);
    check_densified<OptDensHashSketch, float>(800, KMU_ALGO_OPTDENS, KMU_SIG_F32);
}
TEST(test_seq_revoptdensminhash_trait) {
    check_densified<RevOptDensHashSketch, double>(8000, KMU_ALGO_REVOPTDENS, KMU_SIG_F64);
    check_densified<RevOptDensHashSketch, float>(8000, KMU_ALGO_REVOPTDENS, KMU_SIG_F32);
}

// aautils/setsketchert.rs:1394-1466
TEST(test_seqaa_optdensminhash_trait_32bit) {
    SequenceAA seq1 = SequenceAA::from_str(AA1), seq2 = SequenceAA::from_str(AA2);
    std::vector<const SequenceAA *> vseq{&seq1, &seq2};
    SeqSketcherParams sketch_args(5, 80, SketchAlgo::OPTDENS, DataType::AA);
    Ascii a;
    a.add(AA1); a.add(AA2);
    OptDensHashSketch<KmerAA32bit, double> sketcher_f64(sketch_args);
    auto s64 = sketcher_f64.sketch_compressedkmeraa(vseq, kmer_hash_fn);
    CHECK(std::fabs(equal_fraction(s64[0], s64[1]) - 0.5) < 1. / 10.);   // :1444
    CHECK(s64 == oracle_sketch<double>(a, KMU_ALGO_OPTDENS, KMU_KMERAA32BIT, 5, 80, KMU_SIG_F64, KMU_HASHER_NOHASH,
                                       KMU_FHASH_VALUE_MASKED));
    OptDensHashSketch<KmerAA32bit, float> sketcher_f32(sketch_args);
    auto s32 = sketcher_f32.sketch_compressedkmeraa(vseq, kmer_hash_fn);
    CHECK(std::fabs(equal_fraction(s32[0], s32[1]) - 0.5) < 1. / 10.);   // :1465
    CHECK(s32 == oracle_sketch<float>(a, KMU_ALGO_OPTDENS, KMU_KMERAA32BIT, 5, 80, KMU_SIG_F32, KMU_HASHER_NOHASH,
                                      KMU_FHASH_VALUE_MASKED));
}

// HyperLogLogSketch (setsketchert.rs:640-896).  The reference holds no test of it; this one checks what its callers rely on:
// rows in sequence order, one row of hll_params.m registers for a list, registers of a list = maximum over its sequences
// (SetSketcher::merge, :868-885), equal to the oracle's registers
TEST(test_hyperloglog_sketch_trait) {
    std::mt19937_64 rng(640);
    std::vector<Sequence> seqs;
    Ascii a;
    for (size_t len : {size_t(30000), size_t(800), size_t(12000)}) {
        std::string s(len, 'A');
        for (char &c : s) c = "ACGT"[rng() & 3];
        seqs.emplace_back(s);
        a.add(s);
    }
    auto vseq = detail::pointers(seqs);
    SeqSketcherParams seq_params(21, 1024, SketchAlgo::HLL, DataType::DNA);
    SetSketchParams hll_params;
    hll_params.m = 1024;
    HyperLogLogSketch<Kmer64bit, uint16_t> sketcher(seq_params, hll_params, HllSeqsThreading{});
    CHECK(sketcher.get_algo() == SketchAlgo::HLL && sketcher.get_kmer_size() == 21);
    auto per_seq = sketcher.sketch_compressedkmer(vseq, kmer_revcomp_hash_fn);
    auto all = sketcher.sketch_compressedkmer_seqs(vseq, kmer_revcomp_hash_fn);
    CHECK(per_seq.size() == 3 && all.size() == 1 && all[0].size() == 1024);
    for (size_t t = 0; t < 1024; t++) CHECK(all[0][t] == std::max(per_seq[0][t], std::max(per_seq[1][t], per_seq[2][t])));
    kmo_set_hll_params(hll_params.b, hll_params.a, hll_params.q);
    CHECK(per_seq == oracle_sketch<uint16_t>(a, KMU_ALGO_HLL, KMU_KMER64BIT, 21, 1024, KMU_SIG_U16, KMU_HASHER_NOHASH,
                                             KMU_FHASH_CANON_INVHASH));
    HyperLogLogSketch<Kmer64bit, uint32_t> sketcher32(seq_params, hll_params);
    CHECK(sketcher32.sketch_compressedkmer_seqs(vseq, kmer_revcomp_hash_fn) ==
          oracle_sketch<uint32_t>(a, KMU_ALGO_HLL, KMU_KMER64BIT, 21, 1024, KMU_SIG_U32, KMU_HASHER_NOHASH,
                                  KMU_FHASH_CANON_INVHASH, KMU_MODE_ALL_SEQS));
}

// an arbitrary closure (evaluated on the host, sketched on the device) gives what the named closure gives on the device
TEST(test_closure_fallback_equals_device_closure) {
    Sequence seqa(SEQSTR);
    Sequence seqc = seqa.get_reverse_complement();
    std::vector<const Sequence *> vseq{&seqa, &seqc};
    auto canonical_value = [](const Kmer64bit &kmer) -> uint64_t { return kmer.reverse_complement().min(kmer).v; };
    SeqSketcher sk(21, 128);
    CHECK(sk.sketch_probminhash3a<Kmer64bit>(vseq, canonical_value) == sk.sketch_probminhash3a<Kmer64bit>(vseq, FHash::canon_raw));
    auto raw32 = [](const Kmer32bit &kmer) -> uint32_t { return kmer.v; };
    SeqSketcherParams args(7, 32, SketchAlgo::SUPER, DataType::DNA);
    SuperHashSketch<Kmer32bit, double> sup(args);
    CHECK(sup.sketch_compressedkmer(vseq, raw32) == sup.sketch_compressedkmer(vseq, kmer_identity));
    // and a closure no constant names: the k-mer value with its lowest base cleared
    auto odd = [](const Kmer32bit &kmer) -> uint32_t { return kmer.get_compressed_value() & ~3u; };
    auto sig = ProbHash3aSketch<Kmer32bit>(args).sketch_compressedkmer(vseq, odd);
    CHECK(sig.size() == 2 && sig[0].size() == 32);
    for (uint32_t v : sig[0]) CHECK((v & 3u) == 0);
}

// seqblocksketch.rs:459-496
TEST(test_block_32bit_sketch) {
    const std::string seqstra = "TCAAAGGGAAACATTCAAAATCAGTATGCGCCCGTTCAGTTACGTATTGCTCTCGCCGTAGGCCTAATGAGATGGGCTGGGTACAGAG";
    const std::string seqstrb =
        "TCAAAGGGAAATTTTTTTCATTCAAAATCAGTATGCGCCCGTTCAGTTACGTATTGCTCTCGCCGTAGGCCTAATGATTTTTTTGATGGGCTGGGTACAGAG";
    Sequence seqa(seqstra), seqb(seqstrb);
    const size_t block_size = 10, kmer_size = 3, sketch_size = 6;
    BlockSeqSketcher sketcher(block_size, kmer_size, sketch_size);
    BlockSketchedSeq sketcha = sketcher.blocksketch_sequence(1, seqa, kmer_revcomp_hash_fn);
    BlockSketchedSeq sketchb = sketcher.blocksketch_sequence(2, seqb, kmer_revcomp_hash_fn);
    CHECK(sketcha.sketch.size() == (seqstra.size() + block_size - 1) / block_size);   // :108-112
    CHECK(sketchb.sketch.size() == (seqstrb.size() + block_size - 1) / block_size);
    DistBlockSketched mydist;
    CHECK(mydist.eval(sketcha.sketch[0], sketcha.sketch[0]) == 1.f);   // :487
    const float dist_1 = mydist.eval(sketcha.sketch[0], sketchb.sketch[0]);
    const float dist_2 = mydist.eval(sketcha.sketch[1], sketchb.sketch[1]);
    CHECK(dist_1 >= 0.f && dist_1 <= 1.f && dist_2 >= 0.f && dist_2 <= 1.f);
    CHECK(dist_1 < 1.f);   // nine of the first ten 3-mers are shared
    // parity of every block row, both sequences in one call
    kmu_sketch_params p{};
    p.algo = KMU_ALGO_PROB3A; p.kmer_type = KMU_KMER32BIT; p.kmer_size = 3; p.sketch_size = 6; p.sig_type = KMU_SIG_U32;
    p.fhash = KMU_FHASH_CANON_INVHASH; p.block_size = 10;
    Ascii a;
    a.add(seqstra); a.add(seqstrb);
    std::vector<uint64_t> rows(3);
    CHECK(kmu_block_layout(a.off.data(), 2, 10, rows.data()) == 0);
    std::vector<uint32_t> want(rows[2] * 6);
    std::vector<uint8_t> bytes = a.bytes;
    bytes.resize(bytes.size() + 16);
    CHECK(kmo_sketch(&p, bytes.data(), a.off.data(), nullptr, 2, rows.data(), want.data(), nullptr) == 0);
    auto both = sketcher.blocksketch_sequences(1, {&seqa, &seqb}, kmer_revcomp_hash_fn);
    size_t r = 0;
    for (const auto &s : both)
        for (const auto &b : s.sketch) {
            CHECK(b.numseq == s.numseq && b.sketch.size() == 6);
            CHECK(std::equal(b.sketch.begin(), b.sketch.end(), want.begin() + 6 * r));
            r++;
        }
    CHECK(r == rows[2]);
    CHECK(both[0].sketch[3].sketch == sketcha.sketch[3].sketch && both[1].numseq == 2);
}

// minhash.rs:396-435
TEST(test_mininvhash_count_range_intersection) {
    auto vkmer_a = KmerGenerator<Kmer16b32bit>(16).generate_kmer(Sequence(std::string_view(SEQSTR).substr(0, 80)));
    auto vkmer_b = KmerGenerator<Kmer16b32bit>(16).generate_kmer(Sequence(std::string_view(SEQSTR).substr(60)));
    MinInvHashCountKmer<Kmer16b32bit> minhash_a(5), minhash_b(5);
    minhash_a.sketch_kmer_slice(vkmer_a);
    minhash_b.sketch_kmer_slice(vkmer_b);
    auto sketch_a = minhash_a.get_sketchcount();
    auto sketch_b = minhash_b.get_sketchcount();
    CHECK(sketch_a.size() == 5 && sketch_b.size() == 5);
    MinHashDist resdist = mininvhash_distance(minhash_a, minhash_b);
    // the oracle on the same two rows
    std::vector<uint64_t> ha, hb;
    for (auto &e : sketch_a) ha.push_back(e.hashed);
    for (auto &e : sketch_b) hb.push_back(e.hashed);
    CHECK(std::is_sorted(ha.begin(), ha.end()) && std::is_sorted(hb.begin(), hb.end()));
    uint32_t want[3];
    kmo_minhash_distance(ha.data(), 5, hb.data(), 5, want);
    CHECK(resdist.common == want[0] && resdist.total == want[1]);
    // every kept hash is int64_hash of one of the k-mers, and they are the five smallest
    std::vector<uint64_t> all;
    for (auto &k : vkmer_a) all.push_back(kmo_int64_hash(k.v));
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    CHECK(std::equal(ha.begin(), ha.end(), all.begin()));
    for (auto &e : sketch_a) CHECK(e.count == 1);
}

// seqminhash.rs:127-191: bottom-k sketches of two overlapping ranges
TEST(test_minhash_overlapping_ranges) {
    Sequence seq(SEQSTR);
    for (size_t kmer_size : {size_t(16), size_t(10)}) {
        auto sk1 = sketch_seqrange_minhash(seq, {1, 65}, kmer_size, 20);
        auto sk2 = sketch_seqrange_minhash(seq, {35, 75}, kmer_size, 20);
        MinHashDist resdist = minhash_distance(sk1, sk2);
        if (kmer_size == 16) CHECK(resdist.total >= 3);   // :155
        else CHECK(resdist.total == 20);                  // :190
        // the oracle on the same two ranges
        Ascii a;
        a.add(SEQSTR.substr(1, 64)); a.add(SEQSTR.substr(35, 40));
        kmu_sketch_params p{};
        p.algo = KMU_ALGO_BOTTOMK; p.kmer_type = kmer_size == 16 ? KMU_KMER16B32BIT : KMU_KMER32BIT; p.kmer_size = int(kmer_size);
        p.sketch_size = 20; p.sig_type = KMU_SIG_U64; p.hasher = KMU_HASHER_NOHASH; p.fhash = KMU_FHASH_CANON_INVHASH;
        std::vector<uint64_t> h(40);
        std::vector<uint32_t> c(40);
        std::vector<uint8_t> bytes = a.bytes;
        bytes.resize(bytes.size() + 16);
        CHECK(kmo_sketch(&p, bytes.data(), a.off.data(), nullptr, 2, nullptr, h.data(), c.data()) == 0);
        CHECK(sk1.size() == 20 && sk2.size() == 20);
        for (size_t i = 0; i < 20; i++) {
            CHECK(sk1[i].hashed == h[i] && sk1[i].count == c[i]);
            CHECK(sk2[i].hashed == h[20 + i] && sk2[i].count == c[20 + i]);
        }
        uint32_t want[3];
        kmo_minhash_distance(h.data(), 20, h.data() + 20, 20, want);
        CHECK(resdist.common == want[0] && resdist.total == want[1]);
    }
    bool threw = false;
    try {
        sketch_seqrange_minhash(seq, {70, 90}, 16, 20);   // range beyond the sequence: set_range fails, upstream panics
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    CHECK(threw);
}

// seqminhash.rs:195-258: SuperMinHash sketches of the same ranges
TEST(test_superminhash_overlapping_ranges) {
    Sequence seq(SEQSTR);
    struct Case { size_t k, m; double thresh; };
    for (Case cs : {Case{16, 50, 0.15}, Case{10, 20, 0.2}}) {
        auto sk1 = sketch_seqrange_superminhash(seq, {1, 65}, cs.k, cs.m);
        auto sk2 = sketch_seqrange_superminhash(seq, {35, 75}, cs.k, cs.m);
        CHECK(compute_superminhash_jaccard(sk1, sk2) >= cs.thresh);   // :224 / :256
        Ascii a;
        a.add(SEQSTR.substr(1, 64)); a.add(SEQSTR.substr(35, 40));
        auto want = oracle_sketch<double>(a, KMU_ALGO_SUPER, cs.k == 16 ? KMU_KMER16B32BIT : KMU_KMER32BIT, int(cs.k), int(cs.m),
                                          KMU_SIG_F64, KMU_HASHER_NOHASH, KMU_FHASH_CANON_INVHASH);
        CHECK(sk1 == want[0] && sk2 == want[1]);
    }
}

// =========================================================================================================================
// sketching, amino acids (aautils/setsketchert.rs tests)
// =========================================================================================================================

// aautils/setsketchert.rs:1218-1265
TEST(test_seqaa_probminhash_64bit) {
    SequenceAA seq1 = SequenceAA::from_str(AA1), seq2 = SequenceAA::from_str(AA2);
    std::vector<const SequenceAA *> vseq{&seq1, &seq2};
    SeqSketcher sketcher(5, 400);
    auto signatures = sketcher.sketch_probminhash3a<KmerAA64bit>(vseq, kmer_hash_fn);
    const double dist = equal_fraction(signatures[0], signatures[1]);
    CHECK(std::fabs(dist - 0.5) < 1. / 10.);   // :1264
    Ascii a;
    a.add(AA1); a.add(AA2);
    CHECK(signatures == oracle_sketch<uint64_t>(a, KMU_ALGO_PROB3A, KMU_KMERAA64BIT, 5, 400, KMU_SIG_U64, KMU_HASHER_NOHASH,
                                                KMU_FHASH_VALUE_MASKED));
}

// aautils/setsketchert.rs:1268-1317
TEST(test_seqaa_probminhash_trait_64bit) {
    SequenceAA seq1 = SequenceAA::from_str(AA1), seq2 = SequenceAA::from_str(AA2);
    std::vector<const SequenceAA *> vseq{&seq1, &seq2};
    SeqSketcherParams sketch_args(5, 800, SketchAlgo::PROB3A, DataType::AA);
    ProbHash3aSketch<KmerAA64bit> sketcher(sketch_args);
    auto signatures = sketcher.sketch_compressedkmeraa(vseq, kmer_hash_fn);
    CHECK(std::fabs(equal_fraction(signatures[0], signatures[1]) - 0.5) < 1. / 10.);   // :1316
    // the same closure written out, evaluated on the host
    const uint64_t mask = (uint64_t(1) << (AlphabetAA::get_nb_bits() * 5)) - 1;
    auto closure = [mask](const KmerAA64bit &kmer) -> uint64_t { return kmer.get_compressed_value() & mask; };
    CHECK(sketcher.sketch_compressedkmer(vseq, closure) == signatures);
}

// aautils/setsketchert.rs:1320-1391
TEST(test_seqaa_superminhash_trait_64bit) {
    SequenceAA seq1 = SequenceAA::from_str(AA1), seq2 = SequenceAA::from_str(AA2);
    std::vector<const SequenceAA *> vseq{&seq1, &seq2};
    SeqSketcherParams sketch_args(5, 800, SketchAlgo::PROB3A, DataType::AA);
    Ascii a;
    a.add(AA1); a.add(AA2);
    SuperHashSketch<KmerAA64bit, double> sketcher_f64(sketch_args);
    auto s64 = sketcher_f64.sketch_compressedkmeraa(vseq, kmer_hash_fn);
    CHECK(std::fabs(equal_fraction(s64[0], s64[1]) - 0.5) < 1. / 10.);   // :1369
    CHECK(s64 == oracle_sketch<double>(a, KMU_ALGO_SUPER, KMU_KMERAA64BIT, 5, 800, KMU_SIG_F64, KMU_HASHER_NOHASH,
                                       KMU_FHASH_VALUE_MASKED));
    SuperHashSketch<KmerAA64bit, float> sketcher_f32(sketch_args);
    auto s32 = sketcher_f32.sketch_compressedkmeraa(vseq, kmer_hash_fn);
    CHECK(std::fabs(equal_fraction(s32[0], s32[1]) - 0.5) < 1. / 10.);   // :1390
    CHECK(s32 == oracle_sketch<float>(a, KMU_ALGO_SUPER, KMU_KMERAA64BIT, 5, 800, KMU_SIG_F32, KMU_HASHER_NOHASH,
                                      KMU_FHASH_VALUE_MASKED));
}

// aautils/setsketchert.rs:1469-1516
TEST(test_seqaa_probminhash_32bit) {
    SequenceAA seq1 = SequenceAA::from_str(AA1), seq2 = SequenceAA::from_str(AA2);
    std::vector<const SequenceAA *> vseq{&seq1, &seq2};
    SeqSketcher sketcher(5, 400);
    auto signatures = sketcher.sketch_probminhash3a<KmerAA32bit>(vseq, kmer_hash_fn);
    CHECK(std::fabs(equal_fraction(signatures[0], signatures[1]) - 0.5) < 1. / 10.);
    Ascii a;
    a.add(AA1); a.add(AA2);
    CHECK(signatures == oracle_sketch<uint32_t>(a, KMU_ALGO_PROB3A, KMU_KMERAA32BIT, 5, 400, KMU_SIG_U32, KMU_HASHER_NOHASH,
                                                KMU_FHASH_VALUE_MASKED));
    // AA k-mers through the generator: first residue most significant, 5 bits each (kmeraa.rs:301-312)
    auto kmers = KmerGenerator<KmerAA32bit>(5).generate_kmer(seq1);
    CHECK(kmers.size() == AA1.size() - 4);
    KmerAA32bit cur(0, 5);
    for (int i = 0; i < 5; i++) cur = cur.push(uint8_t(AA1[i]));
    CHECK(cur == kmers[0]);
    CHECK(cur.push(uint8_t(AA1[5])) == kmers[1]);
}

// =========================================================================================================================
// counting (kmercount.rs tests)
// =========================================================================================================================

// kmercount.rs:1524-1575
TEST(test_kmer_counter) {
    const size_t nb_random = 1000000;
    KmerCounter<Kmer16b32bit> kmer_counter(0.03, 10000000, 8);
    Sequence seq(SEQSTR);
    std::vector<Kmer16b32bit> vkmer = KmerGenerator<Kmer16b32bit>(16).generate_kmer(seq);
    std::mt19937_64 rng(1524);
    std::uniform_int_distribution<size_t> between(2, vkmer.size() - 1);
    kmer_counter.insert_kmer(vkmer[0]);
    kmer_counter.insert_kmer(vkmer[1]);
    std::map<uint32_t, uint64_t> truth;
    for (size_t i = 0; i < nb_random; i++) {
        const Kmer16b32bit kmer = vkmer[between(rng)];
        kmer_counter.insert_kmer(kmer);
        truth[kmer.v]++;
    }
    kmer_counter.insert_kmer(vkmer[1]);
    CHECK(kmer_counter.get_count(vkmer[0]) == 1);   // :1558
    CHECK(kmer_counter.get_count(vkmer[1]) == 2);   // :1559
    auto countvec = kmer_counter.get_count(vkmer);
    for (size_t i = 2; i < vkmer.size(); i++) CHECK(countvec[i] == std::min<uint64_t>(truth[vkmer[i].v], 255));   // exact, saturating
    CHECK(kmer_counter.get_nb_unique() == 1);
    CHECK(kmer_counter.get_nb_distinct() == vkmer.size());
    kmer_counter.eliminate_once_kmer();   // :110-117: the singleton goes, the others stay as they were
    CHECK(kmer_counter.get_nb_unique() == 0 && kmer_counter.get_count(vkmer[0]) == 0 && kmer_counter.get_count(vkmer[1]) == 2);
    CHECK(kmer_counter.get_nb_distinct() == vkmer.size() - 1);
}

// kmercount.rs:1580-1621
TEST(test_false_positive) {
    KmerCounter<Kmer16b32bit> kmer_counter(0.03, 10000000, 8);
    Sequence seq(SEQSTR);
    std::vector<Kmer16b32bit> vkmer = KmerGenerator<Kmer16b32bit>(16).generate_kmer(seq);
    CHECK(vkmer.size() == 65);   // :1591
    std::mt19937_64 rng(1580);
    std::uniform_int_distribution<size_t> between(vkmer.size() / 2, vkmer.size() - 1);
    for (size_t i = 0; i < 1000000; i++) kmer_counter.insert_kmer(vkmer[between(rng)]);
    auto countvec = kmer_counter.get_count(vkmer);
    for (size_t i = 0; i < countvec.size(); i++) {
        if (i < countvec.size() / 2) CHECK(countvec[i] == 0);   // :1612
        else CHECK(countvec[i] == 255);                          // :1615
    }
}

// count_kmer_threaded_one_to_many (kmercount.rs:881-974) on reads sampled from one genome, against the oracle's counter;
// then the COUNTER_MULTIPLE dump (kmercount.rs:467-531) read back
TEST(test_count_kmer_threaded_one_to_many) {
    std::mt19937_64 rng(881);
    std::string genome(20000, 'A');
    for (char &c : genome) c = "ACGT"[rng() & 3];
    std::vector<Sequence> seqvec;
    Ascii a;
    for (int r = 0; r < 3000; r++) {
        const size_t pos = rng() % (genome.size() - 150);
        std::string read = genome.substr(pos, 150);
        if (rng() & 1) {
            auto rc = Sequence(read).get_reverse_complement().decompress();
            read.assign(rc.begin(), rc.end());
        }
        seqvec.emplace_back(read);
        a.add(read);
    }
    const uint8_t kmer_size = 21;
    auto pool = count_kmer_threaded_one_to_many<Kmer64bit>(seqvec, 4, 8, kmer_size);
    kmu_count_params cp{};
    cp.kmer_type = KMU_KMER64BIT; cp.kmer_size = kmer_size; cp.counter_bits = 8; cp.capacity_hint = 1 << 20;
    kmo_counter *oc = kmo_count_create(&cp);
    std::vector<uint8_t> bytes = a.bytes;
    bytes.resize(bytes.size() + 16);
    CHECK(kmo_count_add_reads(oc, bytes.data(), a.off.data(), uint32_t(seqvec.size())) == 0);
    CHECK(pool->get_nb_distinct() == kmo_count_nb_distinct(oc));
    CHECK(pool->get_nb_unique() == kmo_count_nb_unique(oc));
    uint64_t n = 0;
    kmo_count_dump(oc, 2, nullptr, nullptr, 0, &n);
    std::vector<uint64_t> wk(n);
    std::vector<uint32_t> wc(n);
    kmo_count_dump(oc, 2, wk.data(), wc.data(), n, &n);
    auto [gk, gc] = pool->above2_entries();
    CHECK(gk == wk && gc == wc);
    // queries go through canonical k-mers, as the reference's callers do
    auto kmers = KmerGenerator<Kmer64bit>(kmer_size).generate_kmer(seqvec[0]);
    for (size_t i = 0; i < kmers.size(); i += 13) {
        const Kmer64bit canonical = kmers[i].reverse_complement().min(kmers[i]);
        uint32_t want = 0;
        kmo_count_query(oc, &canonical.v, 1, &want);
        CHECK(pool->get_count(canonical) == want && want >= 1);
        CHECK(pool->get_above2_count(canonical) == (want >= 2 ? want : 0));   // kmercount.rs:100-105
    }
    CHECK(pool->get_count_nb_bits() == 8);
    kmo_count_destroy(oc);
    const std::string fname = "/tmp/kmu_test_mirror.multi_kmer.bin";
    CHECK(pool->dump_kmer_counter(fname) == gk.size());
    std::ifstream in(fname, std::ios::binary);
    uint32_t magic; uint8_t k, nbc; uint64_t nrec;
    in.read(reinterpret_cast<char *>(&magic), 4); in.read(reinterpret_cast<char *>(&k), 1);
    in.read(reinterpret_cast<char *>(&nbc), 1); in.read(reinterpret_cast<char *>(&nrec), 8);
    CHECK(magic == 0xcea2bbff && k == kmer_size && nbc == 1 && nrec == gk.size());
    for (size_t i = 0; i < gk.size(); i++) {
        uint8_t kk, c; uint64_t v;
        in.read(reinterpret_cast<char *>(&kk), 1); in.read(reinterpret_cast<char *>(&v), 8); in.read(reinterpret_cast<char *>(&c), 1);
        CHECK(kk == kmer_size && v == gk[i] && c == std::min<uint32_t>(gc[i], 255));
    }
    std::remove(fname.c_str());
}

// KmerFilter1 / filter1_kmer_16b32bit (kmercount.rs:985-1123): the 16-mers seen exactly once and where they sit
TEST(test_kmer_filter1_once_kmers) {
    std::mt19937_64 rng(985);
    std::string genome(5000, 'A');
    for (char &c : genome) c = "ACGT"[rng() & 3];
    std::vector<Sequence> seqvec;
    Ascii a;
    for (int r = 0; r < 120; r++) {
        const size_t len = 10 + rng() % 300, pos = rng() % (genome.size() - len);
        seqvec.emplace_back(genome.substr(pos, len));
        a.add(genome.substr(pos, len));
    }
    auto filter = filter1_kmer_16b32bit(seqvec);
    kmu_count_params cp{};
    cp.kmer_type = KMU_KMER16B32BIT; cp.kmer_size = 16; cp.counter_bits = 8; cp.capacity_hint = 1 << 18;
    kmo_counter *oc = kmo_count_create(&cp);
    std::vector<uint8_t> bytes = a.bytes;
    bytes.resize(bytes.size() + 16);
    CHECK(kmo_count_add_reads(oc, bytes.data(), a.off.data(), uint32_t(seqvec.size())) == 0);
    uint64_t n = 0;
    CHECK(kmo_count_once_positions(oc, bytes.data(), a.off.data(), uint32_t(seqvec.size()), nullptr, nullptr, nullptr, &n) == 0);
    std::vector<uint64_t> wk(n);
    std::vector<uint32_t> ws(n), wp(n);
    kmo_count_once_positions(oc, bytes.data(), a.off.data(), uint32_t(seqvec.size()), wk.data(), ws.data(), wp.data(), &n);
    CHECK(n > 50 && filter->get_nb_once() == kmo_count_nb_unique(oc));
    kmo_count_destroy(oc);
    auto got = filter->once_positions(detail::gather(detail::pointers(seqvec)));
    CHECK(got.kmin == wk && got.numseq == ws && got.numkmer == wp);
    // every record names a k-mer that really sits there, as its canonical form
    for (size_t i = 0; i < n; i += 7) {
        auto kmers = KmerGenerator<Kmer16b32bit>(16).generate_kmer(seqvec[ws[i]]);
        const Kmer16b32bit kmin = kmers[wp[i]].reverse_complement().min(kmers[wp[i]]);
        CHECK(kmin.v == uint32_t(wk[i]));
    }
    const std::string fname = "/tmp/kmu_test_mirror.once_kmer.bin";
    CHECK(filter->dump_in_file_once_kmer16b32bit(fname, seqvec) == n);
    {   // KmerCountReload (kmercount.rs:1356-1486) reads the dump back
        auto reload = KmerCountReload::load_unique_kmer_from_file(fname);
        CHECK(reload && reload->get_kmer_size() == 16 && reload->get_nb_kmer() == n && reload->kmers().size() == n);
        for (size_t i = 0; i < n; i += 11) {
            auto coord = reload->get_coord_from_rank(i);
            CHECK(coord && coord->read_num == ws[i] && coord->pos == wp[i] && reload->kmers()[i] == uint32_t(wk[i]));
        }
        CHECK(!reload->get_coord_from_rank(n) && !reload->get_multi_kmer_counts());
    }
    {   // the serial drivers count_kmer16b32bit / count_kmer32bit / count_kmer64bit (kmercount.rs:332-362) and the multiple dump
        auto c16 = count_kmer16b32bit(seqvec);
        CHECK(c16->get_nb_unique() == filter->get_nb_once());
        auto pool = count_kmer_thread_independant<Kmer16b32bit>(seqvec, 4, 16);
        CHECK(pool->get_nb_distinct() == c16->get_nb_distinct() && pool->above2_entries() == c16->above2_entries());
        const std::string mname = "/tmp/kmu_test_mirror16.multi_kmer.bin";
        const size_t nm = pool->dump_kmer_counter(mname);
        auto multi = KmerCountReload::load_multiple_kmers_from_file(mname);
        CHECK(multi && multi->get_kmer_size() == 16 && multi->kmers().size() == nm);
        auto counts = multi->get_multi_kmer_counts();
        auto [ek, ec] = pool->above2_entries();
        CHECK(counts && counts->size() == nm);
        for (size_t i = 0; i < nm; i++) CHECK(multi->kmers()[i] == uint32_t(ek[i]) && (*counts)[i] == std::min<uint32_t>(ec[i], 255));
        std::remove(mname.c_str());
        bool threw = false;
        try { count_kmer32bit(seqvec, 15); } catch (const std::invalid_argument &) { threw = true; }
        CHECK(threw);
        threw = false;
        try { count_kmer64bit(seqvec, 16); } catch (const std::invalid_argument &) { threw = true; }
        CHECK(threw);
        CHECK(count_kmer32bit(seqvec, 11)->get_nb_distinct() > 0 && count_kmer64bit(seqvec, 21)->get_nb_distinct() > 0);
    }
    std::ifstream in(fname, std::ios::binary);
    uint32_t magic; uint8_t k; uint64_t nrec;
    in.read(reinterpret_cast<char *>(&magic), 4); in.read(reinterpret_cast<char *>(&k), 1); in.read(reinterpret_cast<char *>(&nrec), 8);
    CHECK(magic == 0xcea2bbdd && k == 16 && nrec == n);
    for (size_t i = 0; i < n; i++) {
        uint32_t rec[3];
        in.read(reinterpret_cast<char *>(rec), 12);
        CHECK(rec[0] == uint32_t(wk[i]) && rec[1] == ws[i] && rec[2] == wp[i]);
    }
    std::remove(fname.c_str());
}

// =========================================================================================================================
// io: the FASTQ reader rule (io.rs:37-57, datasketcher.rs:358-388) and the signature dump
// =========================================================================================================================

TEST(test_parse_fastq_and_signature_dump) {
    std::mt19937_64 rng(358);
    std::string text;
    std::vector<std::string> kept;
    for (int r = 0; r < 200; r++) {
        std::string s(50 + rng() % 400, 'A');
        for (char &c : s) c = "ACGTacgt"[rng() & 7];
        const bool bad = r % 7 == 3;
        if (bad) s[s.size() / 2] = 'N';
        text += "@read" + std::to_string(r) + "\n" + s + (r % 5 == 0 ? "\r\n" : "\n") + "+\n" + std::string(s.size(), 'I') + "\n";
        if (!bad) kept.push_back(s);
    }
    std::vector<uint8_t> bytes(text.begin(), text.end());
    bytes.resize(bytes.size() + 16);
    FastqReads reads = parse_fastq_text(bytes.data(), text.size());
    CHECK(reads.info.n_records == 200 && reads.nb_reads() == kept.size() && reads.info.nb_bad_reads == 200 - kept.size());
    for (size_t i = 0; i < kept.size(); i++)
        CHECK(std::string(reads.bases.begin() + reads.offsets[i], reads.bases.begin() + reads.offsets[i + 1]) == kept[i]);
    // oracle reader on the same text
    std::vector<uint8_t> ob(text.size() + 16);
    std::vector<uint64_t> oo(201);
    std::vector<uint32_t> oi(200);
    uint64_t info[6];
    CHECK(kmo_ingest_fastq(bytes.data(), text.size(), ob.data(), oo.data(), oi.data(), info) == 0);
    CHECK(info[1] == reads.info.n_kept && info[2] == reads.info.kept_bases && info[4] == reads.info.nb_bad_bases);
    // datasketcher's pipeline: sketch the accepted reads, dump, read the file back
    std::vector<Sequence> seqs;
    for (const std::string &s : kept) seqs.emplace_back(s);
    SeqSketcher sketcher(8, 200);
    auto sigs = sketcher.sketch_probminhash3a<Kmer32bit>(detail::pointers(seqs), kmer_revcomp_hash_fn);
    const std::string fname = "/tmp/kmu_test_mirror.sig";
    {
        std::ofstream out = sketcher.create_signature_dump(fname);
        SeqSketcher::dump_signatures_block_u32(sigs, out);
    }
    {
        std::ifstream in(fname, std::ios::binary);
        uint32_t head[4];
        in.read(reinterpret_cast<char *>(head), 16);
        CHECK(head[0] == 0xceabeadd && head[1] == 4 && head[2] == 200 && head[3] == 8);   // seqsketchjaccard.rs:385-414
    }
    SigSketchFileReader reader(fname);   // test_reload_sketch_file, seqsketchjaccard.rs:1015
    CHECK(reader.get_kmer_size() == 8 && reader.get_signature_length() == 200 && reader.get_signature_size() == 4);
    for (const auto &sig : sigs) {
        auto row = reader.next();
        CHECK(row.has_value() && *row == sig);
    }
    CHECK(!reader.next().has_value());
    std::remove(fname.c_str());
    // the parameter dumps next to it (serde_json layout)
    const std::string dir = "/tmp/kmu_test_mirror_json";
    std::filesystem::create_directories(dir);
    SeqSketcherParams(8, 200, SketchAlgo::PROB3A, DataType::DNA).dump_json(dir + "/sketchparams_dump.json");
    {
        std::ifstream in(dir + "/sketchparams_dump.json");
        std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        CHECK(text == "{\"kmer_size\":8,\"sketch_size\":200,\"algo\":\"PROB3A\",\"data_t\":\"DNA\"}");
    }
    SeqSketcherParams back = SeqSketcherParams::reload_json(dir);
    CHECK(back.get_kmer_size() == 8 && back.get_sketch_size() == 200 && back.get_algo() == SketchAlgo::PROB3A && back.get_data_t() == DataType::DNA);
    sketcher.dump_json(dir + "/sketchparams_dump.json");
    SeqSketcher again = SeqSketcher::reload_json(dir);
    CHECK(again.get_kmer_size() == 8 && again.get_sketch_size() == 200);
    std::filesystem::remove_all(dir);
}

// needletail::parse_fastx_file also takes FASTA: multi-line records, format by the first byte
TEST(test_parse_fasta) {
    std::mt19937_64 rng(37);
    std::string text;
    std::vector<std::string> kept;
    for (int r = 0; r < 60; r++) {
        std::string s(rng() % 2000, 'A');
        for (char &c : s) c = "ACGT"[rng() & 3];
        const bool bad = r % 6 == 1 && !s.empty();
        if (bad) s[s.size() / 3] = 'N';
        text += ">contig" + std::to_string(r) + " len=" + std::to_string(s.size()) + "\n";
        for (size_t i = 0; i < s.size(); i += 60) text += s.substr(i, 60) + "\n";
        if (!bad) kept.push_back(s);
    }
    std::vector<uint8_t> bytes(text.begin(), text.end());
    bytes.resize(bytes.size() + 16);
    FastqReads reads = parse_fastx_text(bytes.data(), text.size());
    CHECK(reads.info.n_records == 60 && reads.nb_reads() == kept.size());
    for (size_t i = 0; i < kept.size(); i++)
        CHECK(std::string(reads.bases.begin() + reads.offsets[i], reads.bases.begin() + reads.offsets[i + 1]) == kept[i]);
    std::vector<uint8_t> ob(text.size() + 16);
    std::vector<uint64_t> oo(61);
    uint64_t info[6];
    CHECK(kmo_ingest_fastx(bytes.data(), text.size(), ob.data(), oo.data(), nullptr, info) == 0);
    CHECK(info[0] == 60 && info[1] == reads.info.n_kept && info[2] == reads.info.kept_bases && info[4] == reads.info.nb_bad_bases);
    CHECK(std::equal(ob.begin(), ob.begin() + info[2], reads.bases.begin()));
}

// reads left on the device by the reader (what the two tools use): ranges of them sketch, block-sketch and count to the
// same results as host-resident Sequence objects
TEST(test_device_resident_reads) {
    std::mt19937_64 rng(211);
    std::string text;
    std::vector<Sequence> seqs;
    for (int r = 0; r < 90; r++) {
        std::string s(30 + rng() % 700, 'A');
        for (char &c : s) c = "ACGT"[rng() & 3];
        text += "@r\n" + s + "\n+\n" + std::string(s.size(), 'I') + "\n";
        seqs.emplace_back(s);
    }
    const std::string fname = "/tmp/kmu_test_mirror_dev.fastq";
    { std::ofstream(fname, std::ios::binary) << text; }
    Context &ctx = Context::global();
    DeviceReads reads = DeviceReads::from_file(fname, ctx, 4096);   // small slabs: several uploads
    std::remove(fname.c_str());
    CHECK(reads.nb_reads() == 90 && reads.info.nb_bad_reads == 0);
    auto ptrs = detail::pointers(seqs);
    SeqSketcher sk(8, 64);
    auto want = sk.sketch_probminhash3a<Kmer32bit>(ptrs, kmer_revcomp_hash_fn);
    auto all = sk.sketch_probminhash3a<Kmer32bit>(reads.batch(0, 90), kmer_revcomp_hash_fn);
    CHECK(all == want);
    auto mid = sk.sketch_probminhash3a<Kmer32bit>(reads.batch(37, 61), kmer_revcomp_hash_fn);   // a range, no copy
    CHECK(mid.size() == 24 && std::equal(mid.begin(), mid.end(), want.begin() + 37));
    BlockSeqSketcher bs(100, 8, 16);
    auto wb = bs.blocksketch_sequences(37, {ptrs.begin() + 37, ptrs.begin() + 61}, kmer_revcomp_hash_fn);
    auto gb = bs.blocksketch_sequences(37, reads.batch(37, 61), kmer_revcomp_hash_fn);
    CHECK(gb.size() == wb.size());
    for (size_t i = 0; i < gb.size(); i++) {
        CHECK(gb[i].numseq == wb[i].numseq && gb[i].sketch.size() == wb[i].sketch.size());
        for (size_t j = 0; j < gb[i].sketch.size(); j++) CHECK(gb[i].sketch[j].sketch == wb[i].sketch[j].sketch);
    }
    KmerCounter<Kmer64bit> on_dev(0.03, 1 << 16, 8), on_host(0.03, 1 << 16, 8);
    on_dev.insert_reads(reads.batch(0, 90), 21);
    on_host.insert_reads(ptrs, 21);
    CHECK(on_dev.get_nb_distinct() == on_host.get_nb_distinct() && on_dev.get_nb_unique() == on_host.get_nb_unique());
    CHECK(on_dev.above2_entries() == on_host.above2_entries());
}

// errors surface where the reference panics
TEST(test_errors_where_the_reference_panics) {
    Sequence seqa(SEQSTR);
    std::vector<const Sequence *> vseq{&seqa};
    int status = 0;
    try {
        SeqSketcher(15, 50).sketch_probminhash3a<Kmer32bit>(vseq, kmer_identity);   // k > get_nb_base_max (kmergenerator.rs:48-53)
    } catch (const KmuError &e) {
        status = e.status();
    }
    CHECK(status == KMU_E_BAD_K);
    status = 0;
    try {
        SequenceAA bad("MTEQIELIKLYSTRILAXXAAQ");   // 'X' is not in the alphabet (kmeraa.rs:107)
        SeqSketcher(5, 50).sketch_probminhash3a<KmerAA64bit>(std::vector<const SequenceAA *>{&bad}, kmer_hash_fn);
    } catch (const KmuError &e) {
        status = e.status();
    }
    CHECK(status == KMU_E_BAD_ALPHABET);
}

}  // namespace

int main(int argc, char **argv) {
    int failures = 0, ran = 0;
    for (auto &[name, fn] : registry()) {
        if (argc > 1 && std::find(argv + 1, argv + argc, name) == argv + argc) continue;
        ran++;
        try {
            fn();
            std::printf("ok %s\n", name.c_str());
        } catch (const std::exception &e) {
            failures++;
            std::printf("FAIL %s: %s\n", name.c_str(), e.what());
        }
        std::fflush(stdout);
    }
    std::printf("%d run, %d failed\n", ran, failures);
    return failures;
}
