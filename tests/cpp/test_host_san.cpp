// Host-only code of include/kmerutils.hpp under AddressSanitizer + UndefinedBehaviorSanitizer (`make sanitize`; no GPU, no
// libkmu call): packed sequences, k-mer values, parameter files, the signature dump writer / reader, the k-mer count reloader.
// The reference leans on Rust ownership and bounds checks for these (src/base/sequence.rs, src/sketching/seqsketchjaccard.rs:385-712,
// src/base/kmercount.rs:1148-1503); this side is C++ and gets the sanitizers instead.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "../../include/kmerutils.hpp"

using namespace kmerutils;

static int failures = 0;
#define CHECK(x)                                                        \
    do {                                                                \
        if (!(x)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); failures++; } \
    } while (0)

static std::string random_dna(std::mt19937_64 &rng, size_t n, bool mixed_case) {
    std::string s(n, 'A');
    for (auto &c : s) {
        c = "ACGT"[rng() & 3];
        if (mixed_case && (rng() & 1)) c = char(c + 32);
    }
    return s;
}
static std::string upper(std::string s) {
    for (auto &c : s) c = char(std::toupper((unsigned char) c));
    return s;
}

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    std::mt19937_64 rng(0x5A17);
    // ---- Sequence::new(raw, 2), sequence.rs:25-106: every length around the byte and word edges ----
    for (size_t n : {size_t(0), size_t(1), size_t(3), size_t(4), size_t(5), size_t(15), size_t(16), size_t(17), size_t(63), size_t(64), size_t(65), size_t(1000), size_t(4097)}) {
        const std::string s = random_dna(rng, n, true);
        Sequence q(s);
        CHECK(q.size() == n && q.compressed_length() == (n + 3) / 4);
        const auto d = q.decompress();
        CHECK(std::string(d.begin(), d.end()) == upper(s));
        const auto rr = q.get_reverse_complement().get_reverse_complement().decompress();
        CHECK(std::string(rr.begin(), rr.end()) == upper(s));
        if (n) CHECK(q.get_base(n - 1) == Alphabet2b::encode(uint8_t(s[n - 1])));
    }
    {
        bool threw = false;
        try { Sequence bad(std::string("ACGNT")); } catch (const std::invalid_argument &) { threw = true; }
        CHECK(threw);  // Alphabet2b::encode panics upstream (alphabet.rs:125)
    }
    // ---- k-mer values: push = shift in, reverse_complement an involution, min the canonical form ----
    for (int k = 1; k <= 14; k++) {
        Kmer32bit a = Kmer32bit::build(0, uint8_t(k));
        const std::string s = random_dna(rng, size_t(k) + 7, false);
        for (char c : s) a = a.push(Alphabet2b::encode(uint8_t(c)));
        const auto u = a.get_uncompressed_kmer();
        CHECK(std::string(u.begin(), u.end()) == s.substr(s.size() - size_t(k)));
        CHECK(a.reverse_complement().reverse_complement() == a && a.get_nb_base() == k);
        CHECK(!(a.reverse_complement().min(a) == a) || !(a.reverse_complement() < a));
    }
    {
        Kmer16b32bit a;
        const std::string s = random_dna(rng, 40, false);
        for (char c : s) a = a.push(Alphabet2b::encode(uint8_t(c)));
        const auto u = a.get_uncompressed_kmer();
        CHECK(std::string(u.begin(), u.end()) == s.substr(24) && a.reverse_complement().reverse_complement() == a);
    }
    for (int k = 1; k <= 31; k++) {
        Kmer64bit a = Kmer64bit::build(0, uint8_t(k));
        const std::string s = random_dna(rng, size_t(k) + 11, false);
        for (char c : s) a = a.push(Alphabet2b::encode(uint8_t(c)));
        const auto u = a.get_uncompressed_kmer();
        CHECK(std::string(u.begin(), u.end()) == s.substr(s.size() - size_t(k)));
        CHECK(a.reverse_complement().reverse_complement() == a);
    }
    // ---- parameter files (sketcharg.rs:40-138) ----
    {
        SeqSketcherParams p(21, 400, SketchAlgo(1), DataType(0));
        p.dump_json(dir + "/sketchparams_dump.json");
        const SeqSketcherParams r = SeqSketcherParams::reload_json(dir);
        CHECK(r.get_kmer_size() == 21 && r.get_sketch_size() == 400 && int(r.get_algo()) == 1 && int(r.get_data_t()) == 0);
        bool threw = false;
        try { (void) SeqSketcherParams::reload_json(dir + "/no_such_dir"); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
        std::ofstream(dir + "/sketchparams_dump.json") << "{\"kmer_size\":8";  // truncated: a field is missing
        threw = false;
        try { (void) SeqSketcherParams::reload_json(dir); } catch (const std::exception &) { threw = true; }
        CHECK(threw);
    }
    // ---- the signature dump (seqsketchjaccard.rs:385-414, 572-583) and its reader (:586-712), incl. a truncated last row ----
    {
        const std::string f = dir + "/sigs.bin";
        const size_t m = 37, rows = 11;
        std::vector<std::vector<uint32_t>> sig(rows, std::vector<uint32_t>(m));
        for (auto &r : sig) for (auto &v : r) v = uint32_t(rng());
        {
            std::ofstream out(f, std::ios::binary);
            const uint32_t head[4] = {SeqSketcher::MAGIC_SIG_DUMP, 4u, uint32_t(m), 8u};
            out.write(reinterpret_cast<const char *>(head), sizeof head);
            SeqSketcher::dump_signatures_block_u32(sig, out);
            std::vector<uint32_t> flat;
            for (const auto &r : sig) flat.insert(flat.end(), r.begin(), r.end());
            SeqSketcher::dump_signatures_block_u32(flat, flat.size(), out);
            out.write("xyz", 3);  // a partial row at the end
        }
        SigSketchFileReader rd(f);
        CHECK(rd.get_kmer_size() == 8 && rd.get_signature_length() == m && rd.get_signature_size() == 4);
        size_t n = 0;
        while (auto row = rd.next()) { CHECK(*row == sig[n % rows]); n++; }
        CHECK(n == 2 * rows);
        std::ofstream(f, std::ios::binary).write("\xdd\xea", 2);  // two bytes: no magic
        bool threw = false;
        try { SigSketchFileReader bad(f); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
    }
    // ---- k-mer count dumps (kmercount.rs:1148-1503): both record widths, a truncated record, a wrong magic ----
    for (uint8_t nbc : {uint8_t(1), uint8_t(2)}) {
        const std::string f = dir + "/counts.bin";
        std::vector<std::pair<uint32_t, uint16_t>> recs;
        for (int i = 0; i < 1000; i++) recs.emplace_back(uint32_t(rng()), uint16_t(rng() & (nbc == 1 ? 0xFF : 0xFFFF)));
        {
            std::ofstream out(f, std::ios::binary);
            const uint32_t magic = 0xcea2bbff;
            const uint8_t k = 16;
            const uint64_t n = recs.size();
            out.write(reinterpret_cast<const char *>(&magic), 4); out.write(reinterpret_cast<const char *>(&k), 1);
            out.write(reinterpret_cast<const char *>(&nbc), 1); out.write(reinterpret_cast<const char *>(&n), 8);
            for (auto &r : recs) { out.write(reinterpret_cast<const char *>(&r.first), 4); out.write(reinterpret_cast<const char *>(&r.second), nbc); }
            out.write("\x01\x02", 2);  // half a record
        }
        auto r = KmerCountReload::load_multiple_kmers_from_file(f);
        CHECK(r && r->get_kmer_size() == 16 && r->get_nb_kmer() == recs.size() && r->kmers().size() == recs.size());
        auto c = r ? r->get_multi_kmer_counts() : std::nullopt;
        CHECK(c && c->size() == recs.size() && (*c)[999] == recs[999].second && r->kmers()[0] == recs[0].first);
        CHECK(!r->get_coord_from_rank(0));
        CHECK(KmerCountReload::load_unique_kmer_from_file(f) == nullptr);  // the other dump's magic
        CHECK(KmerCountReload::load_multiple_kmers_from_file(dir + "/missing.bin") == nullptr);
    }
    {
        const std::string f = dir + "/unique.bin";
        {
            std::ofstream out(f, std::ios::binary);
            const uint32_t magic = 0xcea2bbdd;
            const uint8_t k = 16;
            const uint64_t n = 3;
            out.write(reinterpret_cast<const char *>(&magic), 4); out.write(reinterpret_cast<const char *>(&k), 1); out.write(reinterpret_cast<const char *>(&n), 8);
            for (uint32_t i = 0; i < 3; i++) { const uint32_t rec[3] = {100 + i, i, 7 * i}; out.write(reinterpret_cast<const char *>(rec), 12); }
        }
        auto r = KmerCountReload::load_unique_kmer_from_file(f);
        CHECK(r && r->kmers().size() == 3 && !r->get_multi_kmer_counts());
        auto p = r ? r->get_coord_from_rank(2) : std::nullopt;
        CHECK(p && !r->get_coord_from_rank(3));
    }
    std::printf("%s: %d failure(s)\n", argv[0], failures);
    return failures ? 1 : 0;
}
