// diagnostics: where the time of one 10 000-read pack goes in the C++ mirror (alloc / kmu_sketch / download / row split)
#include <chrono>
#include <cstdio>
#include <random>
#include "../../include/kmerutils.hpp"
using namespace kmerutils;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    Context ctx(0);
    std::mt19937_64 rng(1);
    const size_t n = 10000, m = 200;
    std::vector<uint64_t> off(n + 1, 0);
    for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + 3000 + rng() % 6000;
    std::vector<uint8_t> bases(off[n] + 64);
    for (auto &b : bases) b = "ACGT"[rng() & 3];
    DeviceBuffer d_b(ctx, bases.size()), d_o(ctx, off.size() * 8);
    d_b.upload(bases.data(), bases.size());
    d_o.upload(off.data(), off.size() * 8);
    kmu_sketch_params p = detail::sketch_params(KMU_ALGO_PROB3A, KMU_KMER32BIT, 8, m, KMU_SIG_U32, KMU_HASHER_NOHASH,
                                                KMU_FHASH_CANON_INVHASH, 0, KMU_MODE_PER_SEQ, KMU_INPUT_ASCII);
    p.mem = KMU_MEM_DEVICE;
    std::vector<uint32_t> flat(n * m);
    for (int it = 0; it < 4; it++) {
        double t0 = now();
        DeviceBuffer d_sig(ctx, flat.size() * 4);
        double t1 = now();
        ctx.check(kmu_sketch(ctx.raw(), &p, d_b.as<uint8_t>(), d_o.as<uint64_t>(), nullptr, n, nullptr, d_sig.data(), nullptr));
        double t2 = now();
        d_sig.download(flat.data(), flat.size() * 4);
        double t3 = now();
        auto rows = detail::split_rows(flat, n, m);
        double t4 = now();
        d_sig = DeviceBuffer();
        double t5 = now();
        std::printf("alloc %.3f ms, kmu_sketch %.3f ms, download %.3f ms, split %.3f ms, free %.3f ms (%zu bases)\n", (t1 - t0) * 1e3,
                    (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (size_t) off[n]);
    }
    kmu_profile_enable(ctx.raw(), 1);
    ctx.check(kmu_sketch(ctx.raw(), &p, d_b.as<uint8_t>(), d_o.as<uint64_t>(), nullptr, n, nullptr, DeviceBuffer(ctx, flat.size() * 4).data(), nullptr));
    kmu_kernel_stat st[32];
    int k = kmu_profile_get(ctx.raw(), st, 32);
    for (int i = 0; i < k && i < 32; i++) std::printf("  %s: %llu launches, %.3f ms\n", st[i].name, (unsigned long long) st[i].launches, st[i].total_ms);
    return 0;
}
