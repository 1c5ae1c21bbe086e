// Register-persistence probe (PROBES twin only; round 6, probes/contention_regkeep.py).  Question: when several PROCESSES share one GPU and
// the scheduler time-slices their queues (waves are saved and restored by the driver's context-save handler), does a wave get all of its
// registers back?  Each lane fills NR VGPRs with a pattern of (block, thread, register index), keeps adding a per-register constant to
// them for `spin` rounds (plain VALU work, every register live the whole time, no memory), then compares with the closed form and
// records every mismatch as (block, thread, register, got ^ want).  No LDS, no memory traffic inside the loop: anything this kernel
// reports is the platform's, not a kernel's protocol.  Motivation: conv_in's fp32 kernel (csrc/vae_conv.hip: k_conv3x3_fewcin, its inputs
// live in ~80 VGPRs for the whole kernel) returned wrong values for lanes 48-63 of single waves under exactly that load.
#include "common.h"

using namespace mdt;

namespace {
template <int NR>
__global__ __launch_bounds__(256) void k_regkeep(unsigned* __restrict__ out, unsigned* __restrict__ count, unsigned cap, int spin) {
    unsigned r[NR];
    const unsigned tag = (blockIdx.x * 256u + threadIdx.x) * 0x9E3779B1u;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        r[i] = tag + (unsigned)i * 0x85EBCA6Bu;
        asm volatile("" : "+v"(r[i]));
    }
    for (int it = 0; it < spin; ++it) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            r[i] += 2u * (unsigned)i + 1u;
            asm volatile("" : "+v"(r[i]));
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const unsigned want = tag + (unsigned)i * 0x85EBCA6Bu + (unsigned)spin * (2u * (unsigned)i + 1u);
        if (r[i] != want) {
            const unsigned slot = atomicAdd(count, 1u);
            if (slot < cap) {
                out[4 * slot + 0] = blockIdx.x;
                out[4 * slot + 1] = threadIdx.x;
                out[4 * slot + 2] = (unsigned)i;
                out[4 * slot + 3] = r[i] ^ want;
            }
        }
    }
}

// The LDS twin: 18 KB of a known pattern in LDS (what k_conv3x3_fewcin keeps there), read back `spin` times as wave-wide BROADCAST
// ds_read_b128 (every lane the same address) and compared in every lane.
__global__ __launch_bounds__(256) void k_ldskeep(unsigned* __restrict__ out, unsigned* __restrict__ count, unsigned cap, int spin) {
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    constexpr int N = 1152;
    __shared__ u32x4v l4[N];
    for (int i = threadIdx.x; i < N; i += 256) {
        const unsigned b = (unsigned)i * 0x9E3779B1u;
        l4[i] = u32x4v{b, b + 0x85EBCA6Bu, b + 2u * 0x85EBCA6Bu, b + 3u * 0x85EBCA6Bu};
    }
    __syncthreads();
    unsigned bad = 0, first = 0xFFFFFFFFu, fx = 0;
    for (int it = 0; it < spin; ++it) {
#pragma unroll 8
        for (int i = 0; i < N; ++i) {
            const u32x4v v = l4[i];
            const unsigned b = (unsigned)i * 0x9E3779B1u;
            const unsigned x = (v.x ^ b) | (v.y ^ (b + 0x85EBCA6Bu)) | (v.z ^ (b + 2u * 0x85EBCA6Bu)) | (v.w ^ (b + 3u * 0x85EBCA6Bu));
            if (x) {
                ++bad;
                if (first == 0xFFFFFFFFu) { first = (unsigned)i; fx = x; }
            }
        }
    }
    if (bad) {
        const unsigned slot = atomicAdd(count, 1u);
        if (slot < cap) {
            out[4 * slot + 0] = blockIdx.x;
            out[4 * slot + 1] = threadIdx.x;
            out[4 * slot + 2] = first;
            out[4 * slot + 3] = fx;
        }
    }
}
}  // namespace

// d_out: cap x 4 u32 records, d_count: one u32 (zeroed by the caller); nregs in {32, 100, 240}, or 0: the LDS twin (k_ldskeep)
extern "C" int mdtile_probe_regkeep(void* d_out, void* d_count, unsigned cap, int nregs, int spin, int grid_blocks, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_out && d_count && grid_blocks > 0 && spin >= 0, "mdtile_probe_regkeep: bad arguments");
    hipStream_t s = as_stream(stream);
    dim3 g((unsigned)grid_blocks), b(256);
    if (nregs == 32) hipLaunchKernelGGL((k_regkeep<32>), g, b, 0, s, (unsigned*)d_out, (unsigned*)d_count, cap, spin);
    else if (nregs == 100) hipLaunchKernelGGL((k_regkeep<100>), g, b, 0, s, (unsigned*)d_out, (unsigned*)d_count, cap, spin);
    else if (nregs == 240) hipLaunchKernelGGL((k_regkeep<240>), g, b, 0, s, (unsigned*)d_out, (unsigned*)d_count, cap, spin);
    else if (nregs == 0) hipLaunchKernelGGL(k_ldskeep, g, b, 0, s, (unsigned*)d_out, (unsigned*)d_count, cap, spin);
    else MDT_CHECK_ARG(false, "mdtile_probe_regkeep: nregs must be 32, 100, 240, or 0 for the LDS twin");
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}
