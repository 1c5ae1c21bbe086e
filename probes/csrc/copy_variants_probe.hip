// Copy-kernel variants for the blend's floor measurement (PROBES twin of the library only; round 6, probes/blend_r6_ab.py): which plain copy
// is the fastest way to move N bytes in ONE launch on this chip -- plain / write-through stores / non-temporal loads / both, one block per
// 16 KiB or a grid-stride loop.  Result (profiles/r6c): non-temporal loads + write-through stores, any grid: 13.7-13.9 us for 40.1 MB ->
// 40.1 MB cold (5.8 TB/s) against 16.4-17.5 us plain; that form is what mdtile_stream_copy (csrc/blend.hip) ships.
// (The first, restricted LDS-staged blend kernel of round 6 lived in this file as well -- fp32, packed, aligned origins only: 19.1 us cold with
// write-through stores and non-temporal DMA against 19.4 us for k_blend with the same two changes, profiles/r6c -- until the general
// form moved into csrc/blend.hip as k_blend_lds.)
#include "common.h"

using namespace mdt;

// ---- copy-kernel variants for the floor measurement (probes/blend_r6_ab.py): which plain copy is the fastest way to move N bytes in ONE launch?
namespace {
template <int WT, int NT>
__global__ __launch_bounds__(256) void k_copy_var(const uint4* __restrict__ src_, uint4* __restrict__ dst_, size_t n16) {
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    const u32x4v* src = reinterpret_cast<const u32x4v*>(src_);
    u32x4v* dst = reinterpret_cast<u32x4v*>(dst_);
    const size_t stride = (size_t)gridDim.x * 1024;
    for (size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x; base < n16; base += stride) {
        u32x4v v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (base + i * 256 < n16) v[i] = NT ? __builtin_nontemporal_load(src + base + i * 256) : src[base + i * 256];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (base + i * 256 < n16) {
                if (WT) {
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst + base + i * 256), "v"(v[i]) : "memory");
                } else {
                    dst[base + i * 256] = v[i];
                }
            }
    }
}
}  // namespace

// variant bits: 1 = write-through stores, 2 = nontemporal loads; grid_blocks = 0: one block per 16 KiB (no loop), else a grid-stride loop
extern "C" int mdtile_probe_copy(const void* d_src, void* d_dst, size_t bytes, int variant, int grid_blocks, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_src && d_dst && bytes % 16 == 0, "mdtile_probe_copy: bad arguments");
    const size_t n16 = bytes / 16;
    const unsigned g = grid_blocks > 0 ? (unsigned)grid_blocks : (unsigned)((n16 + 1023) / 1024);
    hipStream_t s = as_stream(stream);
    switch (variant & 3) {
        case 0: hipLaunchKernelGGL((k_copy_var<0, 0>), dim3(g), dim3(256), 0, s, (const uint4*)d_src, (uint4*)d_dst, n16); break;
        case 1: hipLaunchKernelGGL((k_copy_var<1, 0>), dim3(g), dim3(256), 0, s, (const uint4*)d_src, (uint4*)d_dst, n16); break;
        case 2: hipLaunchKernelGGL((k_copy_var<0, 1>), dim3(g), dim3(256), 0, s, (const uint4*)d_src, (uint4*)d_dst, n16); break;
        default: hipLaunchKernelGGL((k_copy_var<1, 1>), dim3(g), dim3(256), 0, s, (const uint4*)d_src, (uint4*)d_dst, n16); break;
    }
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}
