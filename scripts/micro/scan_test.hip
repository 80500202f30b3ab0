// micro test: DPP inclusive scan against a serial sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../kmerutils_amd/csrc/kmu_device.h"
__global__ void k(const uint32_t *in, uint32_t *out) { out[threadIdx.x] = kmu::wave_incl_scan_u32(in[threadIdx.x]); }
int main() {
    uint32_t h[128], o[128], *d, *e;
    for (int i = 0; i < 128; i++) h[i] = (uint32_t) (i * 7 + 3) % 11;
    hipMalloc(&d, sizeof h); hipMalloc(&e, sizeof h);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, e);
    hipMemcpy(o, e, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 2; w++) { uint32_t run = 0; for (int i = 0; i < 64; i++) { run += h[w * 64 + i]; if (o[w * 64 + i] != run) { if (bad < 5) printf("lane %d: got %u want %u\n", w*64+i, o[w*64+i], run); bad++; } } }
    printf("scan test: %d mismatches\n", bad);
    return bad != 0;
}
