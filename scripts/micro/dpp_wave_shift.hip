// What do the DPP wave shifts do on gfx950?  (kmu_smer.hip moves per-lane values to the neighbouring lane with them.)
// hipcc --offload-arch=gfx950 -O2 scripts/micro/dpp_wave_shift.hip -o scripts/micro/dpp_wave_shift && scripts/micro/dpp_wave_shift
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    const unsigned v = 100 + threadIdx.x;
    out[threadIdx.x] = (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x130, 0xf, 0xf, false);        // wave_shl:1
    out[64 + threadIdx.x] = (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x138, 0xf, 0xf, false);   // wave_shr:1
    out[128 + threadIdx.x] = (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x134, 0xf, 0xf, false);  // wave_rol:1
    out[192 + threadIdx.x] = (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x101, 0xf, 0xf, false);  // row_shl:1
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = {"wave_shl:1", "wave_shr:1", "wave_rol:1", "row_shl:1"};
    for (int t = 0; t < 4; t++) {
        printf("%s:", names[t]);
        for (int i = 0; i < 64; i++) printf(" %u", h[64 * t + i]);
        printf("\n");
    }
    return 0;
}
