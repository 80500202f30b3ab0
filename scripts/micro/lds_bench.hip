// micro-benchmark (diagnostics, not product): throughput of random-address LDS operations on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 2048
template <int OP, int ILP>
__global__ void __launch_bounds__(1024) k(uint64_t *out, uint32_t S) {
    extern __shared__ uint64_t lds[];
    uint32_t *l32 = (uint32_t *) lds;
    for (uint32_t i = threadIdx.x; i < S; i += blockDim.x) lds[i] = (OP == 1 || OP == 4) ? ~0ull : 0ull;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint64_t acc = 0;
    for (int it = 0; it < N_ITER; it += ILP) {
        uint32_t idx[ILP];
#pragma unroll
        for (int u = 0; u < ILP; u++) { x = x * 1664525u + 1013904223u; idx[u] = (uint32_t) (((uint64_t) (x >> 4) * S) >> 28); }
        uint64_t r[ILP];
#pragma unroll
        for (int u = 0; u < ILP; u++) {
            if (OP == 0) r[u] = *(volatile uint64_t *) &lds[idx[u]];                                  // ds_read_b64
            if (OP == 1) r[u] = atomicCAS((unsigned long long *) &lds[idx[u]], ~0ull, (unsigned long long) (x | 1)); // cmpst rtn (mostly fails after fill)
            if (OP == 2) { atomicAdd(&l32[idx[u]], 1u); r[u] = 0; }                                   // ds_add_u32 no return
            if (OP == 3) r[u] = atomicAdd(&l32[idx[u]], 1u);                                          // ds_add_rtn_u32
            if (OP == 4) r[u] = atomicCAS((unsigned long long *) &lds[idx[u]], 0x1234ull, 0x5678ull); // cmpst rtn, never succeeds
            if (OP == 5) { lds[idx[u]] = x; r[u] = 0; }                                               // ds_write_b64
            if (OP == 6) r[u] = atomicCAS(&l32[idx[u]], 0x1234u, 0x5678u);                            // cmpst_rtn_b32, never succeeds
            if (OP == 7) { atomicAdd((unsigned long long *) &lds[idx[u]], 1ull); r[u] = 0; }          // ds_add_u64 no return
            if (OP == 8) r[u] = atomicAdd((unsigned long long *) &lds[idx[u]], 1ull);                 // ds_add_rtn_u64
            if (OP == 9) r[u] = atomicMin((unsigned long long *) &lds[idx[u]], (unsigned long long) x); // ds_min_rtn_u64
            if (OP == 10) r[u] = atomicOr(&l32[idx[u]], 1u << (x & 31));                              // ds_or_rtn_b32
        }
#pragma unroll
        for (int u = 0; u < ILP; u++) acc += r[u];
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
template <int OP, int ILP>
void run(const char *name, int threads) {
    uint64_t *d; hipMalloc(&d, 64);
    uint32_t S = 8192;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP, ILP><<<256, threads, S * 8>>>(d, S);
    hipEventRecord(a);
    for (int i = 0; i < 5; i++) k<OP, ILP><<<256, threads, S * 8>>>(d, S);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    double ops_per_cu = (double) threads * N_ITER;
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s threads %4d ilp %d : %.3f ms  -> %.2f lane-ops/cycle/CU (%.1f cycles per wave-instr)\n", name, threads, ILP, ms,
           ops_per_cu / cycles, cycles / (ops_per_cu / 64));
    hipFree(d);
}
int main() {
    for (int th : {1024, 256}) {
        if (th == 1024) {
            run<0, 1>("ds_read_b64 random", th); run<0, 4>("ds_read_b64 random", th);
            run<1, 1>("cmpst_rtn_b64 (claims)", th); run<4, 1>("cmpst_rtn_b64 (fails)", th); run<4, 4>("cmpst_rtn_b64 (fails)", th);
            run<2, 1>("ds_add_u32 noret", th); run<2, 4>("ds_add_u32 noret", th);
            run<3, 1>("ds_add_rtn_u32", th); run<3, 4>("ds_add_rtn_u32", th); run<3, 16>("ds_add_rtn_u32", th); run<2, 16>("ds_add_u32 noret", th); run<5, 16>("ds_write_b64", th); run<0, 16>("ds_read_b64 random", th);
            run<5, 1>("ds_write_b64", th); run<5, 4>("ds_write_b64", th);
            run<6, 1>("cmpst_rtn_b32 (fails)", th); run<6, 4>("cmpst_rtn_b32 (fails)", th); run<7, 4>("ds_add_u64 noret", th); run<8, 4>("ds_add_rtn_u64", th);
            run<9, 4>("ds_min_rtn_u64", th); run<10, 4>("ds_or_rtn_b32", th); run<1, 4>("cmpst_rtn_b64 (claims)", th);
        } else {
            run<0, 1>("ds_read_b64 random", th); run<4, 1>("cmpst_rtn_b64 (fails)", th); run<4, 4>("cmpst_rtn_b64 (fails)", th); run<3, 1>("ds_add_rtn_u32", th);
        }
    }
    return 0;
}
