// Standalone timing probe for the blend / gather kernels through the C ABI (no Python, no torch).
//   hipcc --offload-arch=gfx950 -O2 probes/blend_probe.cpp -Iinclude -L<pkg>/mdtile -lmdtile -Wl,-rpath,'$ORIGIN/../<pkg>/mdtile' -o probes/blend_probe
// Prints one line per (workload, method, layout, kernel configuration): average kernel time over back-to-back launches
// (hipEvents on the launch stream) and the algorithmic GB/s of SURVEY.md section 8d.
#include <hip/hip_runtime.h>

#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mdtile.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (float)(x & 0xffff) / 32768.0f - 1.0f;
}
__global__ void k_copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) b[i] = a[i];
}

static float time_launches(hipStream_t s, int iters, const std::function<void()>& fn);

#include <functional>
static float time_launches(hipStream_t s, int iters, const std::function<void()>& fn) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) fn();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3f / iters;  // us per launch
}

struct Work { const char* name; int W, H, tile, ov; };

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int N = 2, C = 4;
    Work works[] = {{"cfg4 8K 128/8", 1024, 1024, 128, 8}, {"cfg4' 8K 128/64", 1024, 1024, 128, 64}, {"cfg3 4K 96/48", 512, 512, 96, 48},
                    {"cfg2 2K 96/48", 256, 256, 96, 48}};
    const char* cfgs[] = {"0,0", "4,4", "8,2", "2,4"};  // 0,0 = the library default heuristic
    // stream-copy reference (64 MiB read + 64 MiB write)
    {
        size_t n = 16u << 20;  // floats
        float *a, *b;
        CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
        k_fill<<<(n + 255) / 256, 256, 0, s>>>(a, n, 1);
        float us = time_launches(s, iters, [&] { k_copy4<<<2048, 256, 0, s>>>((const float4*)a, (float4*)b, n / 4); });
        printf("ref copy 64MiB->64MiB: %.2f us  %.1f GB/s (r+w)\n", us, 2.0 * n * 4 / us * 1e-3);
        CK(hipFree(a)); CK(hipFree(b));
    }
    for (const Work& w : works) {
        mdtile_plan* plan = mdtile_plan_create(w.W, w.H, w.tile, w.tile, w.ov, 4, 1);
        if (!plan) { printf("plan failed: %s\n", mdtile_last_error()); return 1; }
        int info[8];
        mdtile_plan_info(plan, info);
        const int T = info[2], nb = info[3], bs = info[4], tw = info[5], th = info[6];
        const size_t tile_elems = (size_t)tw * th, tiles_elems = (size_t)T * N * C * tile_elems, canvas = (size_t)N * C * w.W * w.H;
        float *d_tiles, *d_x, *d_out, *d_weights, *d_tilew, *d_resc, *d_xt;
        CK(hipMalloc(&d_tiles, tiles_elems * 4)); CK(hipMalloc(&d_xt, tiles_elems * 4));
        CK(hipMalloc(&d_x, canvas * 4)); CK(hipMalloc(&d_out, canvas * 4));
        CK(hipMalloc(&d_weights, (size_t)w.W * w.H * 4)); CK(hipMalloc(&d_resc, (size_t)w.W * w.H * 4)); CK(hipMalloc(&d_tilew, tile_elems * 4));
        k_fill<<<(tiles_elems + 255) / 256, 256, 0, s>>>(d_tiles, tiles_elems, 7);
        k_fill<<<(canvas + 255) / 256, 256, 0, s>>>(d_x, canvas, 9);
        CK(hipMemsetAsync(d_weights, 0, (size_t)w.W * w.H * 4, s));
        mdtile_weight_map_add_grid(plan, nullptr, d_weights, s);
        mdtile_gaussian_weights(tw, th, d_tilew, s);
        CK(hipMemsetAsync(d_resc, 0, (size_t)w.W * w.H * 4, s));
        mdtile_weight_map_add_grid(plan, d_tilew, d_resc, s);
        mdtile_reciprocal(d_resc, d_resc, (size_t)w.W * w.H, s);
        std::vector<const void*> batches(nb);
        for (int b = 0; b < nb; ++b) batches[b] = d_tiles + (size_t)b * bs * N * C * tile_elems;   // same memory viewed as a batch list
        std::vector<void*> xbatches(nb);
        for (int b = 0; b < nb; ++b) xbatches[b] = d_xt + (size_t)b * bs * N * C * tile_elems;
        const double blend_bytes = 4.0 * (tiles_elems + canvas) + 4.0 * w.W * w.H;
        const double gather_bytes = 4.0 * (tiles_elems + canvas);
        float us = time_launches(s, iters, [&] { mdtile_gather_all(plan, MDTILE_DT_F32, N, C, d_x, xbatches.data(), nb, s); });
        printf("%-16s gather_all                     : %8.2f us  %7.1f GB/s   (T=%d, %.1f MB)\n", w.name, us, gather_bytes / us * 1e-3, T, gather_bytes * 1e-6);
        for (int variant = 0; variant < 3; ++variant)   // MD, MoD, MD without the normalising division (debug flag 0x100)
            for (int packed = 1; packed >= 0; --packed) {
                const int method = variant == 1 ? 1 : 0;
                if (!packed && nb > MDTILE_MAX_BATCHES) continue;
                for (const char* cfg : cfgs) {
                    setenv("MDTILE_BLEND_CFG", cfg, 1);
                    mdtile_blend_args a;
                    memset(&a, 0, sizeof(a));
                    a.method = method; a.dtype = MDTILE_DT_F32; a.N = N; a.C = C; a.flags = (packed ? MDTILE_BLEND_PACKED : 0) | (variant == 2 ? 0x100 : 0);
                    a.d_weights = d_weights; a.d_tile_w = d_tilew; a.d_rescale = d_resc; a.d_x_out = d_out;
                    const void* one[1] = {d_tiles};
                    int rc = mdtile_blend(plan, &a, packed ? one : batches.data(), packed ? 1 : nb, nullptr, 0, s);
                    if (rc) { printf("blend failed: %s\n", mdtile_last_error()); return 1; }
                    us = time_launches(s, iters, [&] { mdtile_blend(plan, &a, packed ? one : batches.data(), packed ? 1 : nb, nullptr, 0, s); });
                    const double bytes = blend_bytes + (method ? 4.0 * w.W * w.H : 0.0);
                    printf("%-16s blend %-3s %-6s cfg(PP,G)=%-4s: %8.2f us  %7.1f GB/s  %5.1f%% of 8 TB/s\n", w.name, variant == 1 ? "MoD" : (variant == 2 ? "MDx" : "MD"),
                           packed ? "packed" : "list", cfg, us, bytes / us * 1e-3, bytes / us * 1e-3 / 80.0);
                }
            }
        hipFree(d_tiles); hipFree(d_xt); hipFree(d_x); hipFree(d_out); hipFree(d_weights); hipFree(d_tilew); hipFree(d_resc);
        mdtile_plan_destroy(plan);
    }
    return 0;
}
