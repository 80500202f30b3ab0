// parsefastq -f reads.fastq kmer (--count -s <kmer size> | --unique) [-t threads] [-b 2] [--outdir dir] [--device n]
//
// The counting branch of the reference's tool (src/bin/parsefastq.rs:215-236: `kmer --count`) on the GPU path: the FASTQ
// text is filtered on the device like parse_with_needletail does on the host (src/io.rs:37-57), every canonical k-mer of
// the accepted reads is counted (count_kmer_threaded_one_to_many, src/base/kmercount.rs:881-974) and the k-mers seen at
// least twice are written as a COUNTER_MULTIPLE dump to <fastq>.multi_kmer.bin (kmercount.rs:467-531).  `--unique` is the
// Unicity branch (parsefastq.rs:238-247): the 16-mers seen exactly once, with (sequence, position), to <fastq>.once_kmer.bin.
// Kmer type by size as upstream: k <= 14 Kmer32bit, k == 16 Kmer16b32bit, else Kmer64bit (k <= 31).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/kmerutils.hpp"

using namespace kmerutils;

static void usage() {
    std::fprintf(stderr, "usage: parsefastq -f <fastq> kmer (--count -s <kmer size> | --unique) [-t <threads>] [-b 2] [--outdir <dir>] [--device <n>]\n");
    std::exit(2);
}

template <class Kmer> static void count_and_dump(const DeviceReads &reads, uint8_t kmer_size, const std::string &dumpfname, Context &ctx) {
    KmerCounterPool<Kmer> pool(std::max<uint64_t>(reads.info.kept_bases, 1024), 8, ctx);
    pool.counter().insert_reads(reads.batch(0, reads.nb_reads()), kmer_size);
    std::fprintf(stderr, " nb distinct kmers %llu, nb unique kmers %llu\n", (unsigned long long) pool.get_nb_distinct(),
                 (unsigned long long) pool.get_nb_unique());
    const size_t n = pool.dump_kmer_counter(dumpfname);
    std::fprintf(stderr, " dumped %zu kmers seen at least twice in %s\n", n, dumpfname.c_str());
}

int main(int argc, char **argv) {
    std::string fname, outdir = ".";
    long kmer_size = 0, device = 0;
    bool count = false, kmer_cmd = false, unique = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * {
            if (i + 1 >= argc) usage();
            return argv[++i];
        };
        if (a == "-f" || a == "--file") fname = next();
        else if (a == "kmer") kmer_cmd = true;
        else if (a == "--count") count = true;
        else if (a == "--unique" || a == "-u") unique = true;
        else if (a == "-s" || a == "--size") kmer_size = std::atol(next());
        else if (a == "-t" || a == "--threads") next();   // accepted: the device needs no thread count
        else if (a == "-b" || a == "--bits") {
            if (std::atol(next()) != 2) {
                std::fprintf(stderr, "k-mers need 2 bits per base (kmergenerator.rs:221-223)\n");
                return 2;
            }
        } else if (a == "--outdir") outdir = next();
        else if (a == "--device") device = std::atol(next());
        else usage();
    }
    if (fname.empty() || !kmer_cmd || (!count && !unique) || (count && (kmer_size < 1 || kmer_size > 31))) usage();
    try {
        const auto t0 = std::chrono::steady_clock::now();
        Context ctx{int(device)};
        DeviceReads reads = DeviceReads::from_file(fname, ctx);
        std::fprintf(stderr, " nb reads %llu, nb bases %llu, nb bad bases %llu, nb reads with non acgt %llu\n",
                     (unsigned long long) reads.info.n_records, (unsigned long long) reads.info.n_bases,
                     (unsigned long long) reads.info.nb_bad_bases, (unsigned long long) reads.info.nb_bad_reads);
        // the tool writes <basename>.multi_kmer.bin into the working directory (parsefastq.rs:207-211)
        const size_t slash = fname.rfind('/');
        const std::string dumpfname = outdir + "/" + (slash == std::string::npos ? fname : fname.substr(slash + 1)) + ".multi_kmer.bin";
        if (unique) {   // KmerProcessing::Unicity (parsefastq.rs:238-247): filter1_kmer_16b32bit + the once-k-mer dump
            KmerFilter1 filter(16, uint32_t(std::min<uint64_t>(std::max<uint64_t>(reads.info.kept_bases, 1024), 0xFFFFFFFFu)), ctx);
            detail::Batch all = reads.batch(0, reads.nb_reads());
            filter.insert_reads(all);
            const std::string oncefname = outdir + "/" + (slash == std::string::npos ? fname : fname.substr(slash + 1)) + ".once_kmer.bin";
            std::fprintf(stderr, "dumping unique kmers in file : %s \n", oncefname.c_str());
            const size_t n = filter.dump_in_file_once_kmer16b32bit(oncefname, all);
            std::fprintf(stderr, "dump_in_file_once_kmer16b32bit, number of kmer dumped : %zu (distinct once-k-mers %llu)\n", n,
                         (unsigned long long) filter.get_nb_once());
        } else if (kmer_size <= 14) count_and_dump<Kmer32bit>(reads, uint8_t(kmer_size), dumpfname, ctx);
        else if (kmer_size == 16) count_and_dump<Kmer16b32bit>(reads, uint8_t(kmer_size), dumpfname, ctx);
        else count_and_dump<Kmer64bit>(reads, uint8_t(kmer_size), dumpfname, ctx);
        std::fprintf(stderr, " elapsed time (s) %.3f\n",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "parsefastq: %s\n", e.what());
        return 1;
    }
    return 0;
}
