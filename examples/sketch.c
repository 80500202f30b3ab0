/* The C-ABI from plain C (no C++, no Python, no HIP headers): sketch two sequences with ProbMinHash3a on canonical
 * 16-mers hashed by int32_hash -- the closure of the reference's own tests (src/sketching/seqsketchjaccard.rs:876-880) --
 * and print the fraction of equal slots (what probminhash_get_jaccard_objects reports, :86-108).
 *
 *   make examples/sketch_c && examples/sketch_c
 */
#include <stdio.h>
#include <string.h>

#include "../include/kmu.h"

static const char COMP[256] = {['A'] = 'T', ['C'] = 'G', ['G'] = 'C', ['T'] = 'A'};

int main(void) {
    const char *seqa = "TCAAAGGGAAACATTCAAAATCAGTATGCGCCCGTTCAGTTACGTATTGCTCTCGCTAATGAGATGGGCTGGGTACAGAG";
    const size_t n = strlen(seqa);
    uint8_t bases[256] = {0};
    uint64_t offsets[3] = {0, n, 2 * n};
    memcpy(bases, seqa, n);
    for (size_t i = 0; i < n; i++) bases[n + i] = (uint8_t) COMP[(unsigned char) seqa[n - 1 - i]]; /* reverse complement */

    kmu_ctx *ctx = NULL;
    kmu_device_cfg cfg = {0, 0, NULL, 0};
    if (kmu_create(&cfg, &ctx) != KMU_OK) {
        fprintf(stderr, "kmu_create: %s\n", kmu_last_error(NULL));
        return 1;
    }
    enum { M = 50 };
    kmu_sketch_params p;
    memset(&p, 0, sizeof p);
    p.algo = KMU_ALGO_PROB3A;
    p.kmer_type = KMU_KMER16B32BIT;
    p.kmer_size = 16;
    p.sketch_size = M;
    p.sig_type = KMU_SIG_U32;
    p.hasher = KMU_HASHER_NOHASH;
    p.fhash = KMU_FHASH_CANON_INVHASH;
    p.mode = KMU_MODE_PER_SEQ;
    p.input_kind = KMU_INPUT_ASCII;
    p.mem = KMU_MEM_HOST;
    uint32_t sig[2][M];
    int rc = kmu_sketch(ctx, &p, bases, offsets, NULL, 2, NULL, sig, NULL);
    if (rc != KMU_OK) {
        fprintf(stderr, "kmu_sketch: %s\n", kmu_last_error(ctx));
        kmu_destroy(ctx);
        return 1;
    }
    const uint32_t row_a = 0, row_b = 1; /* rows of the same array */
    uint32_t equal = 0;
    rc = kmu_sig_equal_pairs(ctx, sig, 2, sig, 2, M, KMU_SIG_U32, &row_a, &row_b, 1, KMU_MEM_HOST, &equal);
    if (rc != KMU_OK) {
        fprintf(stderr, "kmu_sig_equal_pairs: %s\n", kmu_last_error(ctx));
        kmu_destroy(ctx);
        return 1;
    }
    printf("jaccard(seqa, revcomp(seqa)) with canonical k-mers = %.3f (expected 1.000)\n", (double) equal / M);
    kmu_destroy(ctx);
    return equal == M ? 0 : 2;
}
