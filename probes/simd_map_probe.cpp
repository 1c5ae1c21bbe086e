// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: [3:0] wave slot, [5:4] SIMD, [11:8] CU, ...)
// The record conv kernels pair the waves of a SIMD for their half-step stagger: csrc/vae_conv_rec.hip / vae_conv_recd.hip assume (w, w + 4).
//   hipcc --offload-arch=gfx950 -O2 -o probes/simd_map_probe probes/simd_map_probe.cpp && probes/simd_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void k(unsigned* out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID, all 32 bits
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
    unsigned* d;
    const int blocks = 512;
    hipMalloc(&d, blocks * 8 * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d);
    static unsigned h[512 * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int pair4 = 0, pair1 = 0, other = 0;
    for (int b = 0; b < blocks; ++b) {
        unsigned simd[8];
        for (int w = 0; w < 8; ++w) simd[w] = (h[b * 8 + w] >> 4) & 3;
        bool p4 = true, p1 = true;
        for (int w = 0; w < 4; ++w) p4 = p4 && simd[w] == simd[w + 4];
        for (int w = 0; w < 8; w += 2) p1 = p1 && simd[w] == simd[w + 1];
        pair4 += p4; pair1 += p1; other += !p4 && !p1;
        if (b < 6) { printf("block %d: SIMD of waves 0..7 =", b); for (int w = 0; w < 8; ++w) printf(" %u", simd[w]); printf("\n"); }
    }
    printf("blocks with partners (w, w+4): %d, (2i, 2i+1): %d, neither: %d of %d\n", pair4, pair1, other, blocks);
    return 0;
}
