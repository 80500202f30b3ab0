// datasketcher -f reads.fastq -k 8 -s 200 -d out.sig [-b block_size] [--device n]
//
// The reference's tool (src/bin/datasketcher.rs:40-312) on the GPU path, without its `ann` sub-command (HNSW is outside
// the path).  The FASTQ / FASTA text is uploaded once, split into records and filtered on the device (a record with a byte outside ACGTacgt is
// dropped and counted, datasketcher.rs:358-388), the accepted reads are sketched in packs with ProbMinHash3a on canonical
// Kmer32bit k-mers hashed by int32_hash (the closure of datasketcher.rs:222-226), whole or by blocks, and the signatures
// are written in the reference's dump formats (seqsketchjaccard.rs:385-414, seqblocksketch.rs:172-226).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/kmerutils.hpp"

using namespace kmerutils;

static void usage() {
    std::fprintf(stderr, "usage: datasketcher -f <fastq> -s <sketch size> -k <kmer size <= 14> -d <dumpfile> [-b <block size>] "
                         "[--device <n>]\n");
    std::exit(2);
}

int main(int argc, char **argv) {
    std::string fname, dumpfname;
    long sketch_size = 0, kmer_size = 0, block_size = 0, device = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * {
            if (i + 1 >= argc) usage();
            return argv[++i];
        };
        if (a == "-f" || a == "--file") fname = next();
        else if (a == "-d" || a == "--dumpfile") dumpfname = next();
        else if (a == "-s" || a == "--sketch") sketch_size = std::atol(next());
        else if (a == "-k" || a == "--kmer") kmer_size = std::atol(next());
        else if (a == "-b" || a == "--block_size") block_size = std::atol(next());
        else if (a == "--device") device = std::atol(next());
        else usage();
    }
    if (fname.empty() || dumpfname.empty() || sketch_size < 2 || kmer_size < 1 || block_size < 0) usage();
    if (kmer_size > long(Kmer32bit::get_nb_base_max())) {
        std::fprintf(stderr, "Kmer32bit holds at most 14 bases (src/base/kmer32bit.rs)\n");
        return 2;
    }
    try {
        const auto t0 = std::chrono::steady_clock::now();
        Context ctx{int(device)};
        DeviceReads reads = DeviceReads::from_file(fname, ctx);   // text up once, accepted reads stay on the device
        const double t_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (reads.info.nb_bad_reads)   // datasketcher.rs:382-384
            std::fprintf(stderr, " number of non acgt sequences %llu \n", (unsigned long long) reads.info.nb_bad_reads);
        const size_t n = reads.nb_reads();
        const size_t sequence_pack = block_size ? 5000 : 10000;   // datasketcher.rs:212
        double t_sketch = 0;
        if (block_size) {
            BlockSeqSketcher sketcher(size_t(block_size), size_t(kmer_size), size_t(sketch_size), ctx);
            std::ofstream out = sketcher.create_signature_dump(dumpfname);
            for (size_t nbseq = 0; nbseq < n; nbseq += sequence_pack) {
                const size_t last = std::min(n, nbseq + sequence_pack);
                BlockSeqSketcher::dump_blocks(out, sketcher.blocksketch_sequences(nbseq, reads.batch(nbseq, last),
                                                                                  kmer_revcomp_hash_fn));
            }
        } else {
            SeqSketcher sketcher(size_t(kmer_size), size_t(sketch_size), ctx);
            std::ofstream out = sketcher.create_signature_dump(dumpfname);
            std::vector<uint32_t> rows;   // one pack of signatures, reused
            for (size_t nbseq = 0; nbseq < n; nbseq += sequence_pack) {
                const size_t last = std::min(n, nbseq + sequence_pack);
                const auto ts = std::chrono::steady_clock::now();
                const size_t nrows = sketcher.sketch_probminhash3a<Kmer32bit>(reads.batch(nbseq, last), kmer_revcomp_hash_fn, rows);
                t_sketch += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
                SeqSketcher::dump_signatures_block_u32(rows, nrows * size_t(sketch_size), out);
            }
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, " nb sequences sketched %zu, elapsed time (s) %.3f (file -> accepted reads on the device %.3f, sketch + dump %.3f of which sketch calls %.3f)\n",
                     n, dt, t_read, dt - t_read, t_sketch);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "datasketcher: %s\n", e.what());
        return 1;
    }
    return 0;
}
