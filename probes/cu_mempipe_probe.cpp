// What one CU's vector-memory pipe sustains on L2-resident data (MI355X): the number every MFMA kernel of this repo is sized
// against since round 4 (DESIGN.md section 3 "The CU's memory pipe").
//   hipcc --offload-arch=gfx950 -O2 probes/cu_mempipe_probe.cpp -o probes/cu_mempipe_probe && probes/cu_mempipe_probe
// Every block owns a CU (persistent-style: grid = number of CUs under test, 64 KB of LDS so that no second block joins it) and
// issues wave-instructions of ONE kind back to back, `U` in flight per wave:
//   dma     global_load_lds_dwordx4   1 KB per wave-instruction, global (L2-resident, shared by all blocks) -> LDS
//   load16  global_load_dwordx4       1 KB per wave-instruction -> VGPRs
//   load4   global_load_dword         256 B per wave-instruction -> VGPRs
//   store16 global_store_dwordx4      1 KB per wave-instruction into a block-private 4 MB stream: one aligned 1 KB run / the same run shifted by
//                                     16 B / two 512 B runs (lanes 0-31, 32-63), aligned or shifted by 16 B = the record conv's epilogue stores
//   store4  global_store_dword        256 B per wave-instruction: one run / two 128 B runs, aligned or shifted = the fp32 epilogue stores
//   mix     dma and load16 alternating (what the attention kernel does: K / Q slabs by DMA, V^T fragments to registers)
// Reported: GB/s per CU and bytes per clock at the 2.4 GHz the chip grants a kernel without MFMA work; also with only 32 CUs
// active (is a limit per CU or chip-wide?) and with 4 instead of 8 waves per CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const char* base, unsigned voff, const u32x4* lds_dst) {
    const unsigned l = (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) u32x4*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(l)
                 : "memory");
}

enum Kind { DMA = 0, LOAD16 = 1, LOAD4 = 2, STORE16 = 3, STORE4 = 4, MIX = 5, STORE16_MIS = 6, STORE16_SPLIT = 7, STORE16_SPLIT_MIS = 8, STORE4_RUNS = 9, STORE4_RUNS_MIS = 10 };

// stores go out through inline asm: a plain C++ store to a loop-invariant address is promoted to a register and sunk out of the loop
__device__ __forceinline__ void st16(char* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st4(char* p, unsigned v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }

// src: `src_bytes` of read-only data shared by every block (L2 / Infinity-Cache resident after the first pass);
// dst: gridDim.x * 4 MB, block-private, written as a stream.  Each wave runs `iters` rounds of U instructions.
template <int KIND, int U>
__global__ __launch_bounds__(1024) void k_pipe(const char* __restrict__ src, unsigned src_bytes, char* __restrict__ dst, int iters, unsigned* sink) {
    __shared__ u32x4 lds[4096];      // 64 KB: one block per CU
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned lane16 = lane * 16, lane4 = lane * 4;
    u32x4 acc = {0u, 0u, 0u, 0u};
    unsigned acc1 = 0;
    char* mine = dst + (size_t)blockIdx.x * (4 << 20);
    // (a cold source -- larger than the 256 MB Infinity Cache -- is walked as a stream: block b, wave w start far apart and advance by nw KB)
    unsigned pos = src_bytes > (64u << 20) ? (unsigned)(((unsigned long long)(blockIdx.x * nw + wave) * 2654435761ull) % (src_bytes >> 10))
                                           : (blockIdx.x * 7919u + wave * 104729u) % (src_bytes >> 10);      // in KB units
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pos = pos + nw;
            if (pos >= (src_bytes >> 10)) pos -= (src_bytes >> 10);
            const char* p = src + ((size_t)pos << 10);
            if (KIND == DMA || (KIND == MIX && (u & 1) == 0)) {
                dma16(p, lane16, lds + (wave * U + u) * 64 % 4096);
            } else if (KIND == LOAD16 || KIND == MIX) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(p + lane16);
                acc ^= v;
            } else if (KIND == LOAD4) {
                acc1 ^= *reinterpret_cast<const unsigned*>(p + lane4);
            } else {
                // streaming destination: every block walks its own 4 MB window (never rewrites a line while it may still sit in L2)
                char* q = mine + ((((size_t)it * U + u) * nw + wave) % 4032) * 1024;
                const u32x4 v = {(unsigned)it, (unsigned)u, (unsigned)lane, (unsigned)wave};
                if (KIND == STORE16) st16(q + lane16, v);                                      // 1 KB contiguous, 1 KB aligned: 8 full lines
                else if (KIND == STORE16_MIS) st16(q + 16 + lane16, v);                        // the same run shifted by one record
                else if (KIND == STORE16_SPLIT) st16(q + (lane & 31) * 16 + (lane >> 5) * (2 << 20), v);        // two aligned 512 B runs 2 MB apart
                else if (KIND == STORE16_SPLIT_MIS) st16(q + 16 + (lane & 31) * 16 + (lane >> 5) * (2 << 20), v);   // ... each shifted by one record (the record epilogue)
                else if (KIND == STORE4) st4(q + lane4, (unsigned)it);                         // 256 B contiguous
                else if (KIND == STORE4_RUNS) st4(q + (lane & 31) * 4 + (lane >> 5) * (2 << 20), (unsigned)it);   // two aligned 128 B runs (the fp32 epilogue)
                else st4(q + 32 + (lane & 31) * 4 + (lane >> 5) * (2 << 20), (unsigned)it);    // ... each shifted by 32 B
            }
        }
        if (KIND == DMA || KIND == MIX) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the round's DMA pieces have landed
    }
    __syncthreads();
    if (KIND == DMA || KIND == MIX) acc ^= lds[threadIdx.x];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w ^ acc1) == 0x12345u) sink[0] = 1;      // keeps the loads alive
}

template <int KIND, int U>
static double run(const char* src, unsigned src_bytes, char* dst, unsigned* sink, int grid, int threads, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_pipe<KIND, U>), dim3(grid), dim3(threads), 0, 0, src, src_bytes, dst, iters / 4, sink);      // warm (caches, clocks)
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_pipe<KIND, U>), dim3(grid), dim3(threads), 0, 0, src, src_bytes, dst, iters, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return best * 1e-3;
}

int main() {
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const unsigned src_bytes = 2u << 20;      // 2 MB: inside every XCD's 4 MB L2
    char *src, *dst;
    unsigned* sink;
    CK(hipMalloc(&src, src_bytes));
    CK(hipMemset(src, 1, src_bytes));
    CK(hipMalloc(&dst, (size_t)cus * (4 << 20) + (4 << 20)));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(sink, 0, 64));
    const char* names[11] = {"dma (global_load_lds_dwordx4)", "load16 (global_load_dwordx4)", "load4 (global_load_dword)", "store16 1 KB run, aligned",
                             "store4 256 B run, aligned", "mix (dma + load16 alternating)", "store16 1 KB run + 16 B", "store16 2 x 512 B runs, aligned",
                             "store16 2 x 512 B runs + 16 B", "store4 2 x 128 B runs, aligned", "store4 2 x 128 B runs + 32 B"};
    const int bytes_per[11] = {1024, 1024, 256, 1024, 256, 1024, 1024, 1024, 1024, 256, 256};
    printf("%s, %d CUs; source: %u KB shared by all blocks (L2-resident); 8 instructions in flight per wave\n", prop.name, cus, src_bytes >> 10);
    printf("%-34s %6s %6s | %10s %10s %12s\n", "kind", "CUs", "waves", "GB/s/CU", "B/clk@2.4", "chip TB/s");
    for (int kind = 0; kind < 11; ++kind) {
        for (int cfg = 0; cfg < 4; ++cfg) {
            const int grid = cfg == 1 ? 32 : cus, threads = cfg == 2 ? 256 : (cfg == 3 ? 1024 : 512);
            const int iters = kind >= 3 && kind != 5 ? 400 : 4000;      // stores: 400 rounds x 8 x 1 KB x 8 waves = 25 MB per CU
            double t = 0;
            switch (kind) {
                case 0: t = run<DMA, 8>(src, src_bytes, dst, sink, grid, threads, iters); break;
                case 1: t = run<LOAD16, 8>(src, src_bytes, dst, sink, grid, threads, iters); break;
                case 2: t = run<LOAD4, 8>(src, src_bytes, dst, sink, grid, threads, iters); break;
                case 3: t = run<STORE16, 8>(src, src_bytes, dst, sink, grid, threads, iters); break;
                case 4: t = run<STORE4, 8>(src, src_bytes, dst, sink, grid, threads, iters); break;
                case 5: t = run<MIX, 8>(src, src_bytes, dst, sink, grid, threads, iters); break;
                case 6: t = run<STORE16_MIS, 8>(src, src_bytes, dst, sink, grid, threads, iters); break;
                case 7: t = run<STORE16_SPLIT, 8>(src, src_bytes, dst, sink, grid, threads, iters); break;
                case 8: t = run<STORE16_SPLIT_MIS, 8>(src, src_bytes, dst, sink, grid, threads, iters); break;
                case 9: t = run<STORE4_RUNS, 8>(src, src_bytes, dst, sink, grid, threads, iters); break;
                default: t = run<STORE4_RUNS_MIS, 8>(src, src_bytes, dst, sink, grid, threads, iters); break;
            }
            const double bytes_cu = (double)(threads / 64) * iters * 8 * bytes_per[kind];
            const double gbs = bytes_cu / t * 1e-9;
            printf("%-34s %6d %6d | %10.1f %10.1f %12.2f\n", names[kind], grid, threads / 64, gbs, gbs / 2.4, gbs * grid * 1e-3);
        }
    }
    // ---- the same loads from a COLD source (2 GB streamed once: every line comes from HBM): what a CU pulls when nothing is resident -- the
    // residual stream of a conv2 epilogue (DESIGN.md section 3 "Late round 4")
    char* cold;
    const unsigned cold_bytes = 2047u << 20;
    CK(hipMalloc(&cold, cold_bytes));
    CK(hipMemset(cold, 1, cold_bytes));
    printf("cold source: %u MB (beyond the Infinity Cache), each line touched once per pass\n", cold_bytes >> 20);
    for (int kind = 0; kind < 3; ++kind)
        for (int cfg = 0; cfg < 2; ++cfg) {
            const int grid = cfg == 1 ? 32 : cus, threads = 512;
            // bytes per block and pass = 8 waves x iters x 8 x bytes_per: keep every pass inside the source without reuse
            const int iters = kind == 2 ? 256 : 64;
            double t = 0;
            switch (kind) {
                case 0: t = run<DMA, 8>(cold, cold_bytes, dst, sink, grid, threads, iters); break;
                case 1: t = run<LOAD16, 8>(cold, cold_bytes, dst, sink, grid, threads, iters); break;
                default: t = run<LOAD4, 8>(cold, cold_bytes, dst, sink, grid, threads, iters); break;
            }
            const double bytes_cu = (double)(threads / 64) * iters * 8 * bytes_per[kind];
            const double gbs = bytes_cu / t * 1e-9;
            printf("cold %-29s %6d %6d | %10.1f %10.1f %12.2f\n", names[kind], grid, threads / 64, gbs, gbs / 2.4, gbs * grid * 1e-3);
        }
    return 0;
}
