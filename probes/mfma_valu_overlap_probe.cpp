// Can the VALU work of one wave hide under the MFMAs of the wave it shares a SIMD with?  (gfx950, 8 waves per block = 2 per SIMD,
// partners (w, w + 4): probes/simd_map_probe.cpp.)  Every wave runs  R x { 12 MFMA 32x32x16 bf16 ; N filler instructions }.
//   mode 0: all waves in the same order (lock step: blocks and fillers coincide on a SIMD)
//   mode 1: waves 4-7 start with the fillers (half a step out of phase with their partners)
//   mode 2: one wave per SIMD only (waves 4-7 idle): what a wave does alone
// filler kind: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 s_nop-free SALU adds, 4 none (MFMA only)
//   hipcc --offload-arch=gfx950 -O2 -o probes/mfma_valu_overlap_probe probes/mfma_valu_overlap_probe.cpp && probes/mfma_valu_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int N>
__device__ __forceinline__ void fillers(float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float& x = f[i & 7];
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
        if (KIND == 1) { f32x2 p = {f[(2 * i) & 7], f[(2 * i + 1) & 7]}; asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p)); f[(2 * i) & 7] = p.x; f[(2 * i + 1) & 7] = p.y; }
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        if (KIND == 3) { int s = i; asm volatile("s_add_u32 %0, %0, 1" : "+s"(s)); }
    }
}

template <int KIND, int N>
__global__ __launch_bounds__(512, 2) void k(float* out, int R, int mode) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int q = 0; q < 16; ++q) acc[a][q] = 0.0f;
    bf16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(0.001f * (threadIdx.x + j)); fb[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
    float f[8];
    for (int j = 0; j < 8; ++j) f[j] = 1.0f + 1e-3f * j;
    if (mode == 2 && wave >= 4) return;
    if (mode == 1 && wave >= 4) fillers<KIND, N>(f);
    for (int r = 0; r < R; ++r) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[a], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        fillers<KIND, N>(f);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.0f;
    for (int a = 0; a < 4; ++a) for (int q = 0; q < 16; ++q) s += acc[a][q];
    for (int j = 0; j < 8; ++j) s += f[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int N>
void run(const char* name, float* d) {
    const int R = 4000, blocks = 256;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(512), 0, 0, d, R, mode);
        hipEventRecord(a);
        hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(512), 0, 0, d, R, mode);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double mf = (mode == 2 ? 4.0 : 8.0) * 12 * R;      // MFMAs per CU
        printf("%-14s N=%3d mode %d: %7.3f ms  -> %6.1f ns per 12-MFMA block per SIMD pair-slot, %5.1f cyc/MFMA/SIMD at 2.4 GHz\n", name, N, mode, ms,
               ms * 1e6 / R, ms * 1e-3 * 2.4e9 / (mf / 4));
    }
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    run<4, 0>("mfma only", d);
    run<0, 48>("v_fma", d);
    run<0, 96>("v_fma", d);
    run<1, 48>("v_pk_fma", d);
    run<2, 24>("v_exp", d);
    run<3, 96>("s_add", d);
    return 0;
}
