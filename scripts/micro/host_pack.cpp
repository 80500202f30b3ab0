// How fast can the GPU box's host cores turn ASCII bases into Sequence::new(raw, 2) bytes (4 bases per byte, first base in bits
// 7..6)?  g++ -O3 -mavx2 -pthread scripts/micro/host_pack.cpp -o scripts/micro/host_pack && scripts/micro/host_pack [GB] [threads...]
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

// 32 bases -> 8 bytes; returns 0 if any byte is outside ACGTacgt
static inline int pack32(const uint8_t *in, uint8_t *out) {
    const __m256i v = _mm256_loadu_si256((const __m256i *) in);
    const __m256i u = _mm256_and_si256(v, _mm256_set1_epi8((char) 0xDF));
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(u, _mm256_set1_epi8('C'))),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(u, _mm256_set1_epi8('T'))));
    // code = x ^ (x >> 1), x = (c >> 1) & 3: A 0, C 1, G 2, T 3
    const __m256i x = _mm256_and_si256(_mm256_srli_epi16(v, 1), _mm256_set1_epi8(3));
    const __m256i c = _mm256_xor_si256(x, _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(1)));
    const __m256i p = _mm256_maddubs_epi16(c, _mm256_set1_epi16(0x0104));          // b0 * 4 + b1 per 16-bit lane
    const __m256i q = _mm256_madd_epi16(p, _mm256_set1_epi32(0x00010010));        // (pair0) * 16 + pair1 per 32-bit lane: one byte
    const __m256i s = _mm256_shuffle_epi8(q, _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                                               0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
    const uint32_t lo = (uint32_t) _mm256_extract_epi32(s, 0), hi = (uint32_t) _mm256_extract_epi32(s, 4);
    memcpy(out, &lo, 4);
    memcpy(out + 4, &hi, 4);
    return _mm256_movemask_epi8(ok) == -1;
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t n = (size_t) (gb * 1e9) & ~(size_t) 31;
    std::vector<uint8_t> in(n), out(n / 4 + 64);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; in[i] = "ACGT"[s & 3]; }
    // correctness of the kernel
    for (size_t i = 0; i < 64; i += 32) {
        uint8_t o[8];
        pack32(&in[i], o);
        for (int b = 0; b < 8; b++) {
            uint8_t w = 0;
            for (int j = 0; j < 4; j++) { const uint8_t ch = in[i + 4 * b + j]; w = (uint8_t) ((w << 2) | (ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3)); }
            if (w != o[b]) { printf("MISMATCH at %zu\n", i + 4 * b); return 1; }
        }
    }
    for (int a = 2; a < std::max(argc, 3); a++) {
        const int T = argc > 2 ? atoi(argv[a]) : 16;
        for (int rep = 0; rep < 3; rep++) {
            std::atomic<size_t> next{0};
            std::atomic<int> bad{0};
            const size_t slab = 4u << 20;
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&] {
                    for (;;) {
                        const size_t b = next.fetch_add(slab);
                        if (b >= n) break;
                        const size_t e = std::min(n, b + slab);
                        int ok = 1;
                        for (size_t i = b; i < e; i += 32) ok &= pack32(&in[i], &out[i / 4]);
                        if (!ok) bad = 1;
                    }
                });
            for (auto &x : th) x.join();
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("threads %3d: %.1f GB/s of bases (%.1f ms for %.2f GB)%s\n", T, n / sec / 1e9, sec * 1e3, n / 1e9, bad ? " BAD" : "");
        }
    }
    return 0;
}
