// Shared helpers for libmdtile (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/mdtile.h"

namespace mdt {

// Probe scaffolding (environment switches that re-shape launches, make kernels skip their stores or hand them a stamp buffer) exists
// only in the PROBES build of the library (mdtile/build.py: build_probes() -> probes/_ab/libmdtile_probes.so, -DMDTILE_PROBES=1; the
// scripts under probes/ ask for it).  The shipping library is compiled with -DMDTILE_PROBES=0: probe_env() is the constant nullptr
// there, so no probe switch is read, none of their names is in the binary, and no launch path calls getenv.  The user-facing
// switches (MDTILE_CONV_MODE, MDTILE_ATTN_MODE, MDTILE_SHARD_TRANSPORT) are read ONCE, when the library is loaded / a context is made.
constexpr bool kProbes = MDTILE_PROBES != 0;
template <bool P = kProbes>
static inline const char* probe_env(const char* name) {
    if constexpr (P) return getenv(name);
    else return nullptr;
}

void set_error(const char* fmt, ...);
bool conv_strict_f32();   // mdtile_set_precision / MDTILE_CONV_MODE=f32: every conv on the exact-fp32 MFMA kernels
bool attn_strict_f32();   // ... / MDTILE_ATTN_MODE=f32: attention on the exact-fp32 kernel
int plan_upload(const struct ::mdtile_plan* plan);  // mirror the plan's lookup tables to the current device (idempotent)

#define MDT_CHECK_ARG(cond, ...)           \
    do {                                   \
        if (!(cond)) {                     \
            mdt::set_error(__VA_ARGS__);   \
            return MDTILE_E_ARG;           \
        }                                  \
    } while (0)

#define MDT_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            mdt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MDTILE_E_HIP;                                                        \
        }                                                                               \
    } while (0)

#define MDT_LAUNCH_CHECK()                                                              \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            mdt::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return MDTILE_E_HIP;                                                        \
        }                                                                               \
    } while (0)

static inline hipStream_t as_stream(mdtile_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// fp32 <-> storage type
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ __hip_bfloat16 from_f32<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

// 4 consecutive elements starting at an address that is only element-aligned (tile origins are arbitrary).
// gfx950 runs in unaligned-access mode, so the f32 case is one global_load_dwordx4.
struct __attribute__((packed, aligned(4))) f32x4_u { float v[4]; };
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&o)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = to_f32<T>(p[j]);
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&o)[4]) {
    f32x4_u t = *reinterpret_cast<const f32x4_u*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = t.v[j];
}
// 16-bit storage: 4 elements = one 8-byte access at 2-byte alignment
struct __attribute__((packed, aligned(2))) u16x4_u { unsigned short v[4]; };
template <> __device__ __forceinline__ void load4<__half>(const __half* p, float (&o)[4]) {
    u16x4_u t = *reinterpret_cast<const u16x4_u*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __half2float(__ushort_as_half(t.v[j]));
}
template <> __device__ __forceinline__ void load4<__hip_bfloat16>(const __hip_bfloat16* p, float (&o)[4]) {
    u16x4_u t = *reinterpret_cast<const u16x4_u*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __uint_as_float((unsigned)t.v[j] << 16);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&o)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = from_f32<T>(o[j]);
}
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&o)[4]) {
    f32x4_u t;
#pragma unroll
    for (int j = 0; j < 4; ++j) t.v[j] = o[j];
    *reinterpret_cast<f32x4_u*>(p) = t;
}

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}

}  // namespace mdt

// The opaque plan (plan.hip owns construction).
struct mdtile_plan {
    int w, h;                 // canvas (latent px)
    int tw, th, ov;           // effective tile size / overlap
    int cols, rows, T;
    int num_batches, tile_bs;
    int* h_table;             // host block [xs | ys | colrange | rowrange | pad | colquad | rowinfo]
    size_t table_len;
    size_t quad_off;          // offset (ints, multiple of 4) of colquad inside the block; rowinfo follows it
    int *h_xs, *h_ys;         // views into h_table: [cols], [rows]
    int *d_xs, *d_ys;         // device mirror (lazy, see mdt::plan_upload)
    int *d_colrange, *d_rowrange;  // per canvas column/row: first covering tile index | (count << 16)
    // 16-byte records for the blend kernel: ONE load gives a thread everything about its columns / its row
    //   colquad[x/4] = { first | count << 16 of the tile columns covering ANY of the 4 px, xs[first], xs[first+1], xs[first+2] }
    //   rowinfo[y]   = { first | count << 16 of the tile rows covering y,               ys[first], ys[first+1], ys[first+2] }
    int4 *d_colquad, *d_rowinfo;
    int nc_max, nr_max;       // most tile columns covering one 4-px quad / most tile rows covering one canvas row (blend kernel dispatch)
};
