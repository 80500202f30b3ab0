// diagnostics: the chip's vector-instruction ISSUE peak, measured -- wave-instructions per cycle per SIMD at 1 / 2 / 4 / 8 waves
// per SIMD with fully independent instructions (8 chains per lane), for the instruction classes the sketch kernels are made of.
// The shader clock is measured too (s_memtime ticks against the 100 MHz wall clock), so the cycle figures do not rest on the
// nominal 2.4 GHz.  bench.py's VALU_ISSUE_PEAK comes from this table (profiles/r03_valu_issue.txt).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int UNROLL = 16; // instructions per chain per loop iteration
constexpr int CHAINS = 8;  // independent chains per lane

// out[block * 2] = s_memtime ticks of wave 0, out[block * 2 + 1] = wall-clock ticks (100 MHz) of the same interval
template <int OP>
__global__ void k_rate(unsigned long long *out, int iters, uint32_t seed) {
    uint32_t a[CHAINS], b = seed | 1u;
    uint64_t q[CHAINS];
    double d[CHAINS];
    for (int c = 0; c < CHAINS; c++) { a[c] = threadIdx.x * 2654435761u + c + seed; q[c] = (uint64_t(a[c]) << 32) | (a[c] * 7u); d[c] = 1.0 + a[c] * 1e-9; }
    double db = 1.0000001;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if constexpr (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                if constexpr (OP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                if constexpr (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                if constexpr (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[c]) : "v"(a[c]), "v"(b) : "vcc");
                if constexpr (OP == 4) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q[c]));
                if constexpr (OP == 5) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[c]) : "v"(db));
                if constexpr (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[c]) : "v"(b) : "vcc");
                if constexpr (OP == 7) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[c]) : "v"(b));
                if constexpr (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[c]) : "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    uint32_t r = 0;
    for (int c = 0; c < CHAINS; c++) r ^= a[c] ^ uint32_t(q[c]) ^ uint32_t(q[c] >> 32) ^ uint32_t(__double_as_longlong(d[c]));
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = w1 - w0; }
    if (r == 0x12345678u) out[0] = r;
}

template <int OP>
int run(const char *name, unsigned long long *out, unsigned long long *h, int cus) {
    const int iters = 20000;
    printf("%-16s", name);
    for (int wps = 1; wps <= 8; wps *= 2) { // waves per SIMD: one workgroup of wps * 256 threads per CU (8: two of 1024)
        const int threads = wps == 8 ? 1024 : wps * 256, blocks = wps == 8 ? 2 * cus : cus;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        k_rate<OP><<<blocks, threads>>>(out, 100, 1);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        k_rate<OP><<<blocks, threads>>>(out, iters, 1);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        CHK(hipMemcpy(h, out, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
        double ticks = 0, wall = 0;
        for (int i = 0; i < blocks; i++) { ticks += (double) h[2 * i]; wall += (double) h[2 * i + 1]; }
        ticks /= blocks;
        wall /= blocks;
        const double winstr = (double) wps * CHAINS * UNROLL * iters; // wave-instructions per SIMD
        // s_memtime ticks per second against the 100 MHz wall clock
        const double tick_hz = ticks / (wall / 1e8);
        const double ghz_evt = winstr / (ms * 1e-3) / 1e9; // wave-instructions per ns per SIMD (from the event time)
        printf(" | %dw: %5.2f G winst/s/SIMD (%5.1f MHz ticks)", wps, ghz_evt, tick_hz / 1e6);
    }
    printf("\n");
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, nominal clock %.0f MHz; rates are wave-instructions per second per SIMD from hipEvent times\n"
           "(2 cycles per wave-instruction at 2.4 GHz = 1.20 G; chip peak = rate x 4 SIMDs x CUs)\n", p.name, cus, p.clockRate / 1000.0);
    unsigned long long *out, *h = new unsigned long long[4 * cus];
    CHK(hipMalloc(&out, sizeof(unsigned long long) * 4 * cus));
    run<0>("v_add_u32", out, h, cus);
    run<1>("v_xor_b32", out, h, cus);
    run<8>("v_fma_f32", out, h, cus);
    run<7>("v_alignbit_b32", out, h, cus);
    run<6>("v_add_co_u32", out, h, cus);
    run<2>("v_mul_lo_u32", out, h, cus);
    run<3>("v_mad_u64_u32", out, h, cus);
    run<4>("v_lshlrev_b64", out, h, cus);
    run<5>("v_fma_f64", out, h, cus);
    CHK(hipFree(out));
    return 0;
}
