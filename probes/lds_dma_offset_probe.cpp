// Does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?  (gfx950)
//   hipcc --offload-arch=gfx950 -O2 probes/lds_dma_offset_probe.cpp -o probes/lds_dma_offset_probe && probes/lds_dma_offset_probe
// One wave, M0 = LDS byte address 0 of an 8 KB buffer pre-filled with 0xFFFFFFFF, source = 8 KB of dwords src[i] = i.
// After `global_load_lds_dwordx4 v(lane * 16), s[src] offset:1024`: which LDS dwords changed, and to what?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const unsigned* src, unsigned* out) {
    __shared__ unsigned lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xFFFFFFFFu;
    __syncthreads();
    const unsigned l = (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) unsigned*)lds;
    const unsigned voff = threadIdx.x * 16;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(l) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}

int main() {
    unsigned *src, *out, h[2048], hs[4096];
    for (int i = 0; i < 4096; ++i) hs[i] = i;
    CK(hipMalloc(&src, sizeof(hs))); CK(hipMalloc(&out, sizeof(h)));
    CK(hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    int first = -1, last = -1;
    for (int i = 0; i < 2048; ++i) if (h[i] != 0xFFFFFFFFu) { if (first < 0) first = i; last = i; }
    printf("LDS dwords written: [%d, %d]  (byte %d .. %d); first value %u (= source dword), i.e. source byte offset %u\n", first, last, first * 4, last * 4 + 3, h[first], h[first] * 4);
    printf("=> LDS destination %s by the immediate offset; global source %s\n", first == 256 ? "MOVES" : (first == 0 ? "does NOT move" : "?"), h[first] == 256 ? "moves" : (h[first] == 0 ? "does not move" : "?"));
    return 0;
}
