// Tile gather (K2) and the gather-formulated overlap blend (K3-K7): ONE launch per model evaluation.
//
// Upstream (tile_methods/multidiffusion.py:147-216, tile_methods/mixtureofdiffusers.py:74-175) scatters: for every
// tile, a strided in-place `x_buffer[slicer] += out_i (* w)` launch, then zero/gt/div/where launches -- ~3T+5 launches
// that each re-read/re-write the canvas.  Here every output pixel is owned by one thread that walks the (<= rows x cols)
// covering tiles in upstream's list order, so
//   * each tile element is read exactly once, the canvas is written exactly once, nothing is zero-filled,
//   * there are no atomics and no races,
//   * the fp32 sum is formed in the same order as upstream's sequential `+=`, i.e. results are bit-identical
//     (this file is compiled with -ffp-contract=off so `a*b + c` stays two roundings, as in eager torch).
// HBM-bound: algorithmic bytes = s*(T*N*C*th*tw + N*C*H*W) + 4*H*W.
//
// Ownership: a thread owns 4 consecutive canvas columns of one row for PP (n,c) planes -> 16 B per lane and plane, 1 KiB
// contiguous per wave and plane on the store side.  The sum over covering tiles is a short in-register loop (1 tile for
// ~75 % of the pixels at overlap 8, at most 9 at overlap > tile / 2); no cross-lane reduction exists in this formulation.
// Two kernels share that ownership and the tail (regions, MD normalisation, feather composite, store):
//   k_blend_lds  (round 6, the default wherever it applies)  every tile ROW SEGMENT that covers the block's strip of one canvas
//                row is copied global -> LDS whole (tw elements = full 16-byte records whatever the tile's canvas origin is) by
//                non-temporal `global_load_lds_dwordx4`, with a 16-byte zero pad on both sides of every staged row; a thread then
//                reads its quad from LDS at tile-relative x.  A tile edge inside a quad, odd tile origins (96 / 48 grids: origins
//                46, 92, ...) and the ragged last quad all take the SAME code path -- pixels outside a tile read the pad's +0.0,
//                which leaves an fp32 sum that started at +0.0 unchanged bit for bit -- where k_blend falls to per-element loads
//                at clamped addresses (4x the load instructions, both branches executed by nearly every wave).
//   k_blend      tile values global -> registers (element-aligned 16-byte loads where the whole quad lies inside the tile):
//                the form of rounds 1-5; now the fallback (tile rows that are not whole 16-byte records, > 3 covering tiles
//                per axis, misaligned batch tensors).
// Both: non-temporal tile loads (every tile value is read exactly once) and write-through (sc0 sc1) canvas stores (the line
// leaves the XCD's L2 with the store instead of staying dirty until the kernel boundary writes it back): 24.6 -> 19.4 us per
// cold 8K evaluation for k_blend alone (profiles/r6c).
#include <type_traits>

#include "common.h"

using namespace mdt;


namespace {

struct BlendParams {
    int W, H, tw, th, cols, tile_bs, N, C;
    int flags, tile_lo, tile_hi, row_lo, nrows, num_regions, num_batches, num_fg;
    int lds_ncs, lds_pad_[3];      // k_blend_lds: most tile columns that touch one strip (the stage is sized for it)
    const int *xs, *ys, *colrange, *rowrange;
    const int4 *colquad, *rowinfo;
    const float *weights, *tile_w, *rescale;
    void* out;
    const float* partial;  // finalize only
    mdtile_region regions[MDTILE_MAX_REGIONS];
    const void* batch[MDTILE_MAX_BATCHES];
};
static_assert(sizeof(BlendParams) <= 4096, "kernel argument block must stay under 4 KiB");

// Shared epilogue: MD normalisation (multidiffusion.py:208) and the foreground feather composite
// (multidiffusion.py:211-216 == mixtureofdiffusers.py:170-175), for one pixel.
template <typename T, int METHOD>
__device__ __forceinline__ float epilogue_px(const BlendParams& P, float acc, int n, int c, int y, int x) {
    float v = acc;
    if (METHOD == MDTILE_METHOD_MD) {
        float w = P.weights[(size_t)y * P.W + x];
        v = w > 1.0f ? acc / w : acc;
    }
    float fbuf = 0.0f, fmask = 0.0f, fcnt = 0.0f;
    for (int k = 0; k < P.num_regions; ++k) {
        const mdtile_region& R = P.regions[k];
        if (R.mode != MDTILE_REGION_FG) continue;
        int ry = y - R.y, rx = x - R.x;
        if (ry < 0 || ry >= R.h || rx < 0 || rx >= R.w) continue;
        size_t off = (size_t)ry * R.w + rx;
        fbuf += to_f32<T>(reinterpret_cast<const T*>(R.out)[((size_t)n * P.C + c) * R.h * R.w + off]);
        fmask += R.weight[off];
        fcnt += 1.0f;
    }
    if (fcnt > 0.0f) {
        if (fcnt > 1.0f) {
            fbuf = fbuf / fcnt;
            fmask = fmask / fcnt;
        }
        v = v * (1.0f - fmask) + fbuf * fmask;
    }
    return v;
}

// One thread owns 4 consecutive canvas columns (a "quad") of one row for PP (n,c) planes.
//   * ONE 16-byte record per axis (plan tables colquad / rowinfo) tells the thread which tile columns / rows cover it and
//     their origins: a single dependent hop before the streaming loads (the old colrange -> xs -> tile chain had three).
//   * the covering tiles are walked in upstream's list order (row-major tile index) in chunks of G candidates; all G*PP
//     16-byte loads of a chunk are issued before the first add, so a lane has up to G*PP KiB-wide wave loads in flight.
//   * plane p = n*C + c of tile t sits at `tile base + p * th*tw` (tile-major batch layout).
// the <= 4 in-canvas floats of a quad, statically indexed (a runtime-indexed local array would be demoted to LDS/scratch)
__device__ __forceinline__ void load_quad_f32(const float* p, int nvalid, float (&o)[4], float dflt) {
    if (nvalid == 4) {
        load4<float>(p, o);
    } else {
        o[0] = p[0];
        o[1] = nvalid > 1 ? p[1] : dflt;
        o[2] = nvalid > 2 ? p[2] : dflt;
        o[3] = dflt;
    }
}

// 4 consecutive elements at an element-aligned address, NON-TEMPORAL (tile values are read once per evaluation)
typedef float f32x4a __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned short u16x4a __attribute__((ext_vector_type(4), aligned(2)));
template <typename T> __device__ __forceinline__ void load4_nt(const T* p, float (&o)[4]);
template <> __device__ __forceinline__ void load4_nt<float>(const float* p, float (&o)[4]) {
    const f32x4a t = __builtin_nontemporal_load(reinterpret_cast<const f32x4a*>(p));
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3];
}
template <> __device__ __forceinline__ void load4_nt<__half>(const __half* p, float (&o)[4]) {
    const u16x4a t = __builtin_nontemporal_load(reinterpret_cast<const u16x4a*>(p));
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __half2float(__ushort_as_half(t[j]));
}
template <> __device__ __forceinline__ void load4_nt<__hip_bfloat16>(const __hip_bfloat16* p, float (&o)[4]) {
    const u16x4a t = __builtin_nontemporal_load(reinterpret_cast<const u16x4a*>(p));
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __uint_as_float((unsigned)t[j] << 16);
}

// 4 consecutive canvas elements, WRITE-THROUGH (sc0 sc1): the line is dropped from the XCD's L2 with the store.  Inline asm: hipcc
// has no builtin for a flat-pointer store with these bits; it pads nothing behind an asm statement, and a store reads its data
// registers after issue -- hence the s_nop 1 (two wait states before a VALU may overwrite them).
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ void store4_wt(T* p, const float (&o)[4]);
template <> __device__ __forceinline__ void store4_wt<float>(float* p, const float (&o)[4]) {
    const f32x4v v = {o[0], o[1], o[2], o[3]};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <> __device__ __forceinline__ void store4_wt<__half>(__half* p, const float (&o)[4]) {
    const u32x2v v = {(unsigned)__half_as_ushort(__float2half_rn(o[0])) | ((unsigned)__half_as_ushort(__float2half_rn(o[1])) << 16),
                      (unsigned)__half_as_ushort(__float2half_rn(o[2])) | ((unsigned)__half_as_ushort(__float2half_rn(o[3])) << 16)};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <> __device__ __forceinline__ void store4_wt<__hip_bfloat16>(__hip_bfloat16* p, const float (&o)[4]) {
    unsigned short h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __hip_bfloat16 b = __float2bfloat16(o[j]);
        h[j] = *reinterpret_cast<const unsigned short*>(&b);
    }
    const u32x2v v = {(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <typename T, bool PACKED>
__device__ __forceinline__ const T* tile_base(const BlendParams& P, int t) {
    const size_t tile_elems = (size_t)P.th * P.tw;
    if (PACKED) return reinterpret_cast<const T*>(P.batch[0]) + (size_t)t * P.N * P.C * tile_elems;
    const int b = t / P.tile_bs, i = t - b * P.tile_bs;   // P.batch[b] with a per-lane b is a (cached) vector load
    return reinterpret_cast<const T*>(P.batch[b]) + (size_t)i * P.N * P.C * tile_elems;
}

// Shared tail of both blend kernels, for one thread's quad x PP planes: background regions (in list order, after every grid tile:
// multidiffusion.py:189-190, mixtureofdiffusers.py:152-153), MD normalisation (:208), foreground feather composite (:191-198, :211-216 ==
// mixtureofdiffusers.py:154-161, :170-175), store.  wq = the MD weight sum of the quad (loaded up front by the caller).
template <typename T, int METHOD, int PP>
__device__ __forceinline__ void blend_tail(const BlendParams& P, float (&acc)[PP][4], const float (&wq)[4], int y, int x0, int p0, int nvalid) {
    // background regions, in list order, after every grid tile (multidiffusion.py:189-190, mixtureofdiffusers.py:152-153)
    for (int k = 0; k < P.num_regions; ++k) {
        const mdtile_region& R = P.regions[k];
        if (R.mode != MDTILE_REGION_BG) continue;
        const int ry = y - R.y;
        if (ry < 0 || ry >= R.h) continue;
        const size_t rplane = (size_t)R.h * R.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rx = x0 + j - R.x;
            if (j >= nvalid || rx < 0 || rx >= R.w) continue;
            const size_t off = (size_t)ry * R.w + rx;
            const T* src = reinterpret_cast<const T*>(R.out) + (size_t)p0 * rplane + off;
            const float wgt = METHOD == MDTILE_METHOD_MOD ? R.weight[off] : 1.0f;
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const float v = to_f32<T>(src[(size_t)pp * rplane]);
                if (METHOD == MDTILE_METHOD_MOD) acc[pp][j] += v * wgt;
                else acc[pp][j] += v;
            }
        }
    }

    const size_t plane_px = (size_t)P.H * P.W;
    const size_t o = ((size_t)p0 * P.H + y) * P.W + x0;
    if (P.flags & MDTILE_BLEND_PARTIAL) {  // raw fp32 sums; the epilogue runs after the cross-rank sum
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            float* dst = reinterpret_cast<float*>(P.out) + o + (size_t)pp * plane_px;
            if (nvalid == 4) store4<float>(dst, acc[pp]);
            else {
                dst[0] = acc[pp][0];
                if (nvalid > 1) dst[1] = acc[pp][1];
                if (nvalid > 2) dst[2] = acc[pp][2];
            }
        }
        return;
    }

    // MD normalisation: x = where(weights > 1, buf / weights, buf)  (multidiffusion.py:208); the weight is per pixel, shared by planes
    if (METHOD == MDTILE_METHOD_MD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (wq[j] > 1.0f) {   // correctly rounded division, only where upstream divides
                const unsigned wb = __float_as_uint(wq[j]);
                if ((wb & 0x007fffffu) == 0u) {
                    // w = 2^k (every overlap count of a plain grid: 2, 4): x / w == x * 2^-k bit for bit (both are the correctly
                    // rounded value of the same exact quotient), and 2^-k is exponent arithmetic -- the 10-instruction IEEE
                    // division sequence x 32 values per thread sat between the last load and the first store of every wave
                    const float r = __uint_as_float(0x7f000000u - wb);
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp) acc[pp][j] = acc[pp][j] * r;
                } else {
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp) acc[pp][j] = acc[pp][j] / wq[j];
                }
            }
        }
    }

    // foreground feather composite (multidiffusion.py:191-198, 211-216 == mixtureofdiffusers.py:154-161, 170-175)
    if (P.num_fg > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= nvalid) continue;
            float fbuf[PP], fmask = 0.0f, fcnt = 0.0f;
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) fbuf[pp] = 0.0f;
            for (int k = 0; k < P.num_regions; ++k) {
                const mdtile_region& R = P.regions[k];
                if (R.mode != MDTILE_REGION_FG) continue;
                const int ry = y - R.y, rx = x0 + j - R.x;
                if (ry < 0 || ry >= R.h || rx < 0 || rx >= R.w) continue;
                const size_t rplane = (size_t)R.h * R.w, off = (size_t)ry * R.w + rx;
                const T* src = reinterpret_cast<const T*>(R.out) + (size_t)p0 * rplane + off;
#pragma unroll
                for (int pp = 0; pp < PP; ++pp) fbuf[pp] += to_f32<T>(src[(size_t)pp * rplane]);
                fmask += R.weight[off];
                fcnt += 1.0f;
            }
            if (fcnt > 0.0f) {
                if (fcnt > 1.0f) fmask = fmask / fcnt;
#pragma unroll
                for (int pp = 0; pp < PP; ++pp) {
                    float fb = fbuf[pp];
                    if (fcnt > 1.0f) fb = fb / fcnt;
                    acc[pp][j] = acc[pp][j] * (1.0f - fmask) + fb * fmask;
                }
            }
        }
    }

#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        T* dst = reinterpret_cast<T*>(P.out) + o + (size_t)pp * plane_px;
        if (nvalid == 4) store4_wt<T>(dst, acc[pp]);
        else {
            dst[0] = from_f32<T>(acc[pp][0]);
            if (nvalid > 1) dst[1] = from_f32<T>(acc[pp][1]);
            if (nvalid > 2) dst[2] = from_f32<T>(acc[pp][2]);
        }
    }
}

template <typename T, int METHOD, int PP, int G, bool PACKED>
__global__ __launch_bounds__(256) void k_blend(const BlendParams P) {
    const int W4 = (P.W + 3) >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W4 * P.nrows) return;
    const int yq = idx / W4, xq = idx - yq * W4;
    const int y = P.row_lo + yq;
    const int x0 = xq << 2;
    const int p0 = blockIdx.y * PP;                        // the host guarantees N*C % PP == 0
    const int nvalid = P.W - x0 < 4 ? P.W - x0 : 4;
    const size_t tile_elems = (size_t)P.th * P.tw;

    float acc[PP][4];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[pp][j] = 0.f;

    // MD: the per-pixel weight sum for the epilogue, requested up front so its latency hides behind the tile loads
    float wq[4] = {1.f, 1.f, 1.f, 1.f};
    if (METHOD == MDTILE_METHOD_MD && !(P.flags & MDTILE_BLEND_PARTIAL)) load_quad_f32(P.weights + (size_t)y * P.W + x0, nvalid, wq, 1.0f);

    if (P.num_batches > 0) {
        const int4 cq = P.colquad[xq];
        const int4 rq = P.rowinfo[y];
        const int c0 = cq.x & 0xffff, nc = cq.x >> 16, r0 = rq.x & 0xffff, nr = rq.x >> 16;
        // "clean" quad: all 4 px in the canvas and every candidate column contains the whole quad -> 16-byte loads.
        // True for every quad when the tile origins are multiples of 4.  Other quads (a tile edge inside the quad, the
        // ragged last quad of a row) take the same chunked walk with per-element loads at clamped addresses and a
        // per-pixel coverage mask.  Only > 3 covering tiles per axis (overlap > 2/3 of the tile) falls to the generic walk.
        // (Measured: folding both kinds of load into one walk is slower on clean grids -- more code per candidate.)
        const bool small = nc <= 3 && nr <= 3;
        bool clean = nvalid == 4 && small;
        {
            const int t0 = x0 - cq.y, t1 = x0 - cq.z, t2 = x0 - cq.w;
            clean = clean && t0 >= 0 && t0 + 3 < P.tw;
            if (nc > 1) clean = clean && t1 >= 0 && t1 + 3 < P.tw;
            if (nc > 2) clean = clean && t2 >= 0 && t2 + 3 < P.tw;
        }
        auto chunked_walk = [&](auto vec_tag) {
            constexpr bool VEC = decltype(vec_tag)::value;
            const int total = nr * nc;
            float resc[4] = {0.f, 0.f, 0.f, 0.f};
            if (METHOD == MDTILE_METHOD_MOD) load_quad_f32(P.rescale + (size_t)y * P.W + x0, nvalid, resc, 0.f);
            int rr = 0, cc = 0;  // running candidate, row-major == ascending tile index == upstream's list order
            for (int s0 = 0; s0 < total; s0 += G) {
                float v[G][PP][4], wg[G][4];
                unsigned cov[G];      // bit j: pixel j of the quad is covered by candidate g (0xf for every live clean candidate)
                const T* row[G];      // tile row start (+ tx for VEC)
                size_t woff[G];       // same position inside the [th, tw] tile-weight map
                int txs[G];
                // phase A: addresses of the chunk's candidates (batch-pointer lookups, if any, all issued together)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int xo = cc == 0 ? cq.y : (cc == 1 ? cq.z : cq.w);
                    const int yo = rr == 0 ? rq.y : (rr == 1 ? rq.z : rq.w);
                    const int t = (r0 + rr) * P.cols + c0 + cc;
                    bool ok = s0 + g < total;
                    if (P.flags & MDTILE_BLEND_TILE_RANGE) ok = ok && t >= P.tile_lo && t < P.tile_hi;
                    const int tx = x0 - xo;
                    unsigned m = 0xfu;
                    if (!VEC) {
                        m = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < nvalid && tx + j >= 0 && tx + j < P.tw) m |= 1u << j;
                    }
                    cov[g] = ok ? m : 0u;
                    txs[g] = tx;
                    woff[g] = (size_t)(y - yo) * P.tw + (VEC ? tx : 0);
                    row[g] = tile_base<T, PACKED>(P, cov[g] ? t : 0) + (size_t)p0 * tile_elems + woff[g];
                    ++cc;
                    if (cc == nc) { cc = 0; ++rr; }
                }
                // phase B: every load of the chunk in flight before the first add
#pragma unroll
                for (int g = 0; g < G; ++g) {
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp)
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[g][pp][j] = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) wg[g][j] = 0.f;
                    if (cov[g]) {
                        if (VEC) {                 // whole quad inside the tile: 16-byte loads
#pragma unroll
                            for (int pp = 0; pp < PP; ++pp) load4_nt<T>(row[g] + (size_t)pp * tile_elems, v[g][pp]);
                            if (METHOD == MDTILE_METHOD_MOD) load4<float>(P.tile_w + woff[g], wg[g]);
                        } else {                   // a tile edge inside the quad / ragged last quad: per-element loads
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                int cx = txs[g] + j;          // clamped into the tile row: always a valid address
                                cx = cx < 0 ? 0 : (cx >= P.tw ? P.tw - 1 : cx);
#pragma unroll
                                for (int pp = 0; pp < PP; ++pp) v[g][pp][j] = to_f32<T>(row[g][(size_t)pp * tile_elems + cx]);
                                if (METHOD == MDTILE_METHOD_MOD) wg[g][j] = P.tile_w[woff[g] + cx];
                            }
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (!cov[g]) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!VEC && !((cov[g] >> j) & 1u)) continue;
                        if (METHOD == MDTILE_METHOD_MOD) {
                            const float w = wg[g][j] * resc[j];              // w = tile_weights * rescale_factor[slicer]  (mixtureofdiffusers.py:125)
#pragma unroll
                            for (int pp = 0; pp < PP; ++pp) acc[pp][j] += v[g][pp][j] * w;   // x_buffer[slicer] += out * w  (:126)
                        } else {
#pragma unroll
                            for (int pp = 0; pp < PP; ++pp) acc[pp][j] += v[g][pp][j];       // multidiffusion.py:167
                        }
                    }
                }
            }
        };
        if (clean) {
            chunked_walk(std::true_type{});
        } else if (small) {
            chunked_walk(std::false_type{});
        } else {
            // generic per-pixel walk through the colrange / rowrange / xs / ys tables
#pragma unroll 1
            for (int j = 0; j < nvalid; ++j) {
                const int x = x0 + j;
                const int cr = P.colrange[x], rrg = P.rowrange[y];
                const int pc0 = cr & 0xffff, pnc = cr >> 16, pr0 = rrg & 0xffff, pnr = rrg >> 16;
                float a[PP];
#pragma unroll
                for (int pp = 0; pp < PP; ++pp) a[pp] = 0.f;
                for (int r = pr0; r < pr0 + pnr; ++r) {
                    const int ty = y - P.ys[r];
                    for (int c = pc0; c < pc0 + pnc; ++c) {
                        const int t = r * P.cols + c;
                        if ((P.flags & MDTILE_BLEND_TILE_RANGE) && (t < P.tile_lo || t >= P.tile_hi)) continue;
                        const size_t toff = (size_t)ty * P.tw + (x - P.xs[c]);
                        const T* src = tile_base<T, PACKED>(P, t) + (size_t)p0 * tile_elems + toff;
                        float wgt = 1.0f;
                        if (METHOD == MDTILE_METHOD_MOD) wgt = P.tile_w[toff] * P.rescale[(size_t)y * P.W + x];
#pragma unroll
                        for (int pp = 0; pp < PP; ++pp) {
                            const float v = to_f32<T>(src[(size_t)pp * tile_elems]);
                            if (METHOD == MDTILE_METHOD_MOD) a[pp] += v * wgt;
                            else a[pp] += v;
                        }
                    }
                }
#pragma unroll
                for (int pp = 0; pp < PP; ++pp) {  // acc[pp][j] = a[pp] without dynamic register indexing
                    if (j == 0) acc[pp][0] = a[pp];
                    else if (j == 1) acc[pp][1] = a[pp];
                    else if (j == 2) acc[pp][2] = a[pp];
                    else acc[pp][3] = a[pp];
                }
            }
        }
    }

    blend_tail<T, METHOD, PP>(P, acc, wq, y, x0, p0, nvalid);
}

// ---- the LDS-staged form (see the file header) ------------------------------------------------------------------------
// Block = `blockDim.x` quads (a strip of 4 * blockDim.x px) of ONE canvas row x LPP planes.  LDS: stage[rr][p][cl] = one tile row each,
// [16 B of zeros | tw elements | 16 B of zeros], rr = covering tile row (<= 3), p = plane, cl = tile column relative to the first one
// that touches the strip (P.lds_ncs of them at most; the host sizes the stage for it).  One wave-instruction of LDS-DMA moves one staged
// row (tw * sizeof(T) / 16 lanes active: the LDS destination of `global_load_lds` is wave-uniform base + 16 * lane, the global source
// is per lane).  A 4-element read at tile-relative x in [-3, tw - 1] then never leaves [pad | row | pad].
template <typename T, int METHOD, int LPP, bool PACKED>
__global__ __launch_bounds__(256) void k_blend_lds(const BlendParams P) {
    extern __shared__ uint4 lds16[];
    constexpr int PADE = 16 / (int)sizeof(T);                 // pad elements on each side of a staged row
    const int W4 = (P.W + 3) >> 2, SQ = blockDim.x, strips = (W4 + SQ - 1) / SQ;
    const int yq = blockIdx.x / strips, strip = blockIdx.x - yq * strips;
    const int y = P.row_lo + yq;
    const int p0 = blockIdx.y * LPP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = SQ >> 6;
    const int pitch = P.tw + 2 * PADE;                        // elements per staged row
    const int pitch16 = pitch * (int)sizeof(T) / 16;          // ... in 16-byte records (tw * sizeof(T) % 16 == 0: host check)
    const size_t tile_elems = (size_t)P.th * P.tw;
    T* const stage = reinterpret_cast<T*>(lds16);

    // strip-uniform geometry (the same values in every lane; made scalar for the DMA addressing)
    const int xq_lo = strip * SQ, xq_hi = min(xq_lo + SQ - 1, W4 - 1);
    const int cqlo = __builtin_amdgcn_readfirstlane(P.colquad[xq_lo].x), cqhi = __builtin_amdgcn_readfirstlane(P.colquad[xq_hi].x);
    const int4 rq = P.rowinfo[y];
    const int c_lo = cqlo & 0xffff, ncs = (cqhi & 0xffff) + (cqhi >> 16) - c_lo;
    const int r0 = __builtin_amdgcn_readfirstlane(rq.x) & 0xffff, nr = __builtin_amdgcn_readfirstlane(rq.x) >> 16;
    const int yo3[3] = {__builtin_amdgcn_readfirstlane(rq.y), __builtin_amdgcn_readfirstlane(rq.z), __builtin_amdgcn_readfirstlane(rq.w)};
    const int rows_total = nr * LPP * ncs;

    // ---- pads: 2 x 16 bytes of zeros per staged row (the DMA never writes them)
    for (int i = tid; i < 2 * rows_total; i += SQ) lds16[(i >> 1) * pitch16 + ((i & 1) ? pitch16 - 1 : 0)] = make_uint4(0u, 0u, 0u, 0u);
    // ---- stage: one wave-instruction per (tile row, plane, tile column).  The (tile row, plane) pairs are dealt to the waves; a wave walks
    // the tile columns of its pairs with running addresses (no division on this path: it is the block's critical one)
    const int lanes_row = P.tw * (int)sizeof(T) / 16;
    for (int q = wave; q < nr * LPP; q += nwaves) {
        const int rr = q / LPP, p = q - rr * LPP;              // (LPP is a power of two)
        const int yo = rr == 0 ? yo3[0] : (rr == 1 ? yo3[1] : yo3[2]);
        const size_t in_tile = (size_t)(p0 + p) * tile_elems + (size_t)(y - yo) * P.tw;
        int t = (r0 + rr) * P.cols + c_lo;
        uint4* dst = lds16 + (size_t)q * ncs * pitch16 + 1;
        for (int cl = 0; cl < ncs; ++cl, ++t, dst += pitch16) {
            if ((P.flags & MDTILE_BLEND_TILE_RANGE) && (t < P.tile_lo || t >= P.tile_hi)) continue;   // not this rank's tile: may not even exist
            const T* src = tile_base<T, PACKED>(P, t) + in_tile;
            if (lane < lanes_row)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const uint4*>(src) + lane),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 2 /* nt */);
        }
    }
    // ---- per-thread state that does not depend on the staged data: requested before the barrier
    const int xq = xq_lo + tid;
    const bool live = xq < W4;
    const int x0 = xq << 2;
    const int nvalid = P.W - x0 < 4 ? P.W - x0 : 4;
    float wq[4] = {1.f, 1.f, 1.f, 1.f}, resc[4] = {0.f, 0.f, 0.f, 0.f};
    int4 cq = make_int4(0, 0, 0, 0);
    if (live) {
        cq = P.colquad[xq];
        if (METHOD == MDTILE_METHOD_MD && !(P.flags & MDTILE_BLEND_PARTIAL)) load_quad_f32(P.weights + (size_t)y * P.W + x0, nvalid, wq, 1.0f);
        if (METHOD == MDTILE_METHOD_MOD) load_quad_f32(P.rescale + (size_t)y * P.W + x0, nvalid, resc, 0.f);
    }
    __syncthreads();      // hipcc puts vmcnt(0) in front of it while an LDS-DMA is pending: this wave's rows have landed, the barrier covers the others'
    if (!live) return;

    float acc[LPP][4];
#pragma unroll
    for (int pp = 0; pp < LPP; ++pp)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[pp][j] = 0.f;
    const int c0 = cq.x & 0xffff, nc = cq.x >> 16;
    const int plane_stride = ncs * pitch;                     // elements between the staged rows of two planes
    for (int rr = 0; rr < nr; ++rr) {                          // candidates row-major == ascending tile index == upstream's list order
        const int yo = rr == 0 ? yo3[0] : (rr == 1 ? yo3[1] : yo3[2]);
        for (int cc = 0; cc < nc; ++cc) {
            const int xo = cc == 0 ? cq.y : (cc == 1 ? cq.z : cq.w);
            const int tx = x0 - xo;                            // in [-3, tw - 1]: the candidate covers at least one pixel of the quad
            const int t = (r0 + rr) * P.cols + c0 + cc;
            const bool ok = !(P.flags & MDTILE_BLEND_TILE_RANGE) || (t >= P.tile_lo && t < P.tile_hi);
            const T* s = stage + (size_t)(rr * LPP * ncs + (c0 + cc - c_lo)) * pitch + PADE + tx;
            float wg[4] = {0.f, 0.f, 0.f, 0.f};
            if (METHOD == MDTILE_METHOD_MOD) {
                // w = tile_weights * rescale_factor[slicer] (mixtureofdiffusers.py:125) for the covered pixels, 0 for the others
                const float* wrow = P.tile_w + (size_t)(y - yo) * P.tw;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cx = tx + j;
                    const bool cov = ok && j < nvalid && cx >= 0 && cx < P.tw;
                    const float w = wrow[cx < 0 ? 0 : (cx >= P.tw ? P.tw - 1 : cx)] * resc[j];
                    wg[j] = cov ? w : 0.f;
                }
            }
#pragma unroll
            for (int pp = 0; pp < LPP; ++pp) {
                float v[4];
                load4<T>(s + (size_t)pp * plane_stride, v);   // element-aligned LDS read (ds_read2_b32 pairs: 8 LDS cycles per wave -- not what bounds this kernel)
                if (!ok) v[0] = v[1] = v[2] = v[3] = 0.f;     // a tile that was not staged: whatever the LDS holds there
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (METHOD == MDTILE_METHOD_MOD) acc[pp][j] += v[j] * wg[j];      // x_buffer[slicer] += out * w  (:126); uncovered: + 0 * 0
                    else acc[pp][j] += v[j];                                          // multidiffusion.py:167; uncovered: + 0.0 from the pad
                }
            }
        }
    }
    blend_tail<T, METHOD, LPP>(P, acc, wq, y, x0, p0, nvalid);
}

template <typename T, int METHOD>
__global__ __launch_bounds__(256) void k_blend_finalize(const BlendParams P) {
    const size_t plane_px = (size_t)P.nrows * P.W;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= plane_px) return;
    const int yq = (int)(idx / P.W);
    const int y = P.row_lo + yq, x = (int)(idx - (size_t)yq * P.W);
    const int plane = blockIdx.y, n = plane / P.C, c = plane - n * P.C;
    const size_t o = (((size_t)n * P.C + c) * P.H + y) * P.W + x;
    reinterpret_cast<T*>(P.out)[o] = from_f32<T>(epilogue_px<T, METHOD>(P, P.partial[o], n, c, y, x));
}

// ---- gather ------------------------------------------------------------------------------------------------------
struct GatherParams {
    int W, H, tw, th, cols, tile_bs, N, C;
    int t_lo, t_hi, packed, _pad;
    const int *xs, *ys;
    const void* x_in;
    void* batch[MDTILE_MAX_BATCHES];
};
static_assert(sizeof(GatherParams) <= 4096, "kernel argument block must stay under 4 KiB");

// grid: x = chunks of 4 columns over (th rows x tw4), y = plane (n*C+c), z = tile (t_lo + z)
template <typename T>
__global__ __launch_bounds__(256) void k_gather(const GatherParams P) {
    const int tw4 = (P.tw + 3) >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= tw4 * P.th) return;
    const int ty = idx / tw4, tx0 = (idx - ty * tw4) << 2;
    const int plane = blockIdx.y, n = plane / P.C, c = plane - n * P.C;
    const int t = P.t_lo + blockIdx.z;
    const int r = t / P.cols, cc = t - r * P.cols;
    const T* src = reinterpret_cast<const T*>(P.x_in) + (((size_t)n * P.C + c) * P.H + P.ys[r] + ty) * P.W + P.xs[cc] + tx0;
    T* dst;
    const size_t tile_elems = (size_t)P.th * P.tw;
    if (P.packed) {
        dst = reinterpret_cast<T*>(P.batch[0]) + (((size_t)t * P.N + n) * P.C + c) * tile_elems;
    } else {
        int b = t / P.tile_bs, i = t - b * P.tile_bs;
        dst = reinterpret_cast<T*>(P.batch[b]) + (((size_t)i * P.N + n) * P.C + c) * tile_elems;
    }
    dst += (size_t)ty * P.tw + tx0;
    const int nvalid = P.tw - tx0 < 4 ? P.tw - tx0 : 4;
    if (nvalid == 4) {
        float v[4];
        load4<T>(src, v);
        store4<T>(dst, v);
    } else {
        for (int j = 0; j < nvalid; ++j) dst[j] = src[j];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_gather_rect(const T* __restrict__ x_in, T* __restrict__ out, int C, int W, int H,
                                                     int x0, int y0, int w, int h) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= w * h) return;
    const int y = idx / w, x = idx - y * w;
    const int plane = blockIdx.y;
    out[(size_t)plane * w * h + idx] = x_in[((size_t)plane * H + y0 + y) * W + x0 + x];
}

// the blend's measurement floor (mdtile_stream_copy): every thread moves 4 x 16 bytes, a wave 4 x 1 KiB runs a block-stride apart;
// non-temporal loads + write-through (sc0 sc1) stores -- the fastest of the copy forms probes/blend_r6_ab.py times for a single cold
// ~80 MB launch (profiles/r6c: 13.9 us = 5.8 TB/s against 16.9 us with plain loads and stores; HIP's uint4 struct type instead of the
// ext-vector type: 36 us)
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_copy(const u32x4v* __restrict__ src, u32x4v* __restrict__ dst, size_t n16) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    u32x4v v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + i * 256 < n16) v[i] = __builtin_nontemporal_load(src + base + i * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + i * 256 < n16) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst + base + i * 256), "v"(v[i]) : "memory");
}

// (planes per thread, candidates per chunk).  PP must divide N*C; MDTILE_BLEND_CFG="PP,G" overrides the default (probing).
template <typename T, int PP, int G>
void launch_blend_cfg(const BlendParams& P, int method, hipStream_t s) {
    dim3 grid(cdiv((long long)P.nrows * ((P.W + 3) / 4), 256), (P.N * P.C) / PP), block(256);
    const bool packed = (P.flags & MDTILE_BLEND_PACKED) != 0;
    if (method == MDTILE_METHOD_MD) {
        if (packed) hipLaunchKernelGGL((k_blend<T, MDTILE_METHOD_MD, PP, G, true>), grid, block, 0, s, P);
        else hipLaunchKernelGGL((k_blend<T, MDTILE_METHOD_MD, PP, G, false>), grid, block, 0, s, P);
    } else {
        if (packed) hipLaunchKernelGGL((k_blend<T, MDTILE_METHOD_MOD, PP, G, true>), grid, block, 0, s, P);
        else hipLaunchKernelGGL((k_blend<T, MDTILE_METHOD_MOD, PP, G, false>), grid, block, 0, s, P);
    }
}

// k_blend_lds: does it take this launch, and in what shape?
//   WHERE IT PAYS (profiles/r6e, cold, same process): grids whose tile origins are not all multiples of 4 -- the 96 / 48 class, upstream's
//   default tile settings -- where k_blend's edge quads take per-element loads and nearly every wave runs both branches: 4096^2
//   Mixture-of-Diffusers 24.3 -> 15.6 us, 2048^2 MultiDiffusion 9.5 -> 8.0 us.  On grids with origins % 4 == 0 the two kernels tie
//   (8K 128 / 8: 20.0-21.4 vs 20.0-20.8 us) or k_blend wins (8K 128 / 64, 15 tile columns per strip: 34-38 vs 30 us): those stay on k_blend.
//   WHAT IT NEEDS: a grid (num_batches > 0); tile rows that are whole 16-byte records of at most one wave-instruction (tw * sizeof(T) % 16
//   == 0, <= 1 KiB) at 16-byte aligned addresses; at most 3 covering tiles per axis (the plan records carry three origins); a stage of
//   <= 64 KB per block with LPP in {4, 2, 1} planes per block.  Blocks are always 256 threads (4 waves issue the DMA) even when the canvas
//   row has fewer quads: 18.2 -> 15.6 us at W = 512.
// MDTILE_BLEND_LDS = "0" (off) | "1" (wherever it can run) | "SQ,LPP" (strip quads, planes per block) in the PROBES twin.
template <typename T, int LPP>
void launch_blend_lds_cfg(const BlendParams& P, int method, int SQ, size_t lds, hipStream_t s) {
    const int W4 = (P.W + 3) / 4;
    dim3 grid((unsigned)(((W4 + SQ - 1) / SQ) * P.nrows), (P.N * P.C) / LPP), block(SQ);
    const bool packed = (P.flags & MDTILE_BLEND_PACKED) != 0;
    if (method == MDTILE_METHOD_MD) {
        if (packed) hipLaunchKernelGGL((k_blend_lds<T, MDTILE_METHOD_MD, LPP, true>), grid, block, lds, s, P);
        else hipLaunchKernelGGL((k_blend_lds<T, MDTILE_METHOD_MD, LPP, false>), grid, block, lds, s, P);
    } else {
        if (packed) hipLaunchKernelGGL((k_blend_lds<T, MDTILE_METHOD_MOD, LPP, true>), grid, block, lds, s, P);
        else hipLaunchKernelGGL((k_blend_lds<T, MDTILE_METHOD_MOD, LPP, false>), grid, block, lds, s, P);
    }
}

template <typename T>
bool launch_blend_lds(BlendParams& P, const mdtile_plan* plan, int method, hipStream_t s) {
    const int row_bytes = P.tw * (int)sizeof(T);
    if (P.num_batches <= 0 || row_bytes % 16 != 0 || row_bytes > 1024 || plan->nc_max > 3 || plan->nr_max > 3 || plan->nc_max < 1) return false;
    const int nptr = (P.flags & MDTILE_BLEND_PACKED) ? 1 : P.num_batches;
    for (int b = 0; b < nptr; ++b)
        if (P.batch[b] && ((uintptr_t)P.batch[b] & 15)) return false;
    int SQ = 256, lpp_forced = 0;
    bool wanted = false;
    for (int c = 0; c < plan->cols; ++c) wanted = wanted || (plan->h_xs[c] % 4 != 0);
    if (const char* e = probe_env("MDTILE_BLEND_LDS")) {
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 64 && a <= 256 && a % 64 == 0) { SQ = a; lpp_forced = b; wanted = true; }
        else wanted = atoi(e) != 0;
    }
    if (!wanted) return false;
    int ncs = 1;      // most tile columns that touch one strip of 4 * SQ px
    for (int sx = 0; sx < P.W; sx += 4 * SQ) {
        int n = 0;
        for (int c = 0; c < plan->cols; ++c) n += (plan->h_xs[c] < sx + 4 * SQ && plan->h_xs[c] + P.tw > sx) ? 1 : 0;
        ncs = n > ncs ? n : ncs;
    }
    const int planes = P.N * P.C;
    const size_t per_plane = (size_t)plan->nr_max * ncs * (row_bytes + 32);
    int lpp = 0;
    for (int cand : {4, 2, 1})
        if (planes % cand == 0 && per_plane * cand <= 64 * 1024 && (lpp_forced == 0 || lpp_forced == cand)) { lpp = cand; break; }
    if (lpp == 0) return false;
    P.lds_ncs = ncs;
    const size_t lds = per_plane * lpp;
    if (lpp == 4) launch_blend_lds_cfg<T, 4>(P, method, SQ, lds, s);
    else if (lpp == 2) launch_blend_lds_cfg<T, 2>(P, method, SQ, lds, s);
    else launch_blend_lds_cfg<T, 1>(P, method, SQ, lds, s);
    return true;
}

template <typename T>
int launch_blend(BlendParams& P, const mdtile_plan* plan, int method, bool finalize, hipStream_t s) {
    dim3 block(256);
    if (finalize) {
        dim3 grid(cdiv((long long)P.nrows * P.W, 256), P.N * P.C);
        if (method == MDTILE_METHOD_MD) hipLaunchKernelGGL((k_blend_finalize<T, MDTILE_METHOD_MD>), grid, block, 0, s, P);
        else hipLaunchKernelGGL((k_blend_finalize<T, MDTILE_METHOD_MOD>), grid, block, 0, s, P);
    } else if (plan && launch_blend_lds<T>(P, plan, method, s)) {
        // (the LDS-staged kernel took it)
    } else {
        const int planes = P.N * P.C;
        // planes per thread: as many as keep >= ~128k threads in the grid (2 per lane of the chip), then G so that a
        // thread has ~16 16-byte loads in flight (measured on MI355X: (8,2) for the 8K canvas, (4,4)/(2,4) below it)
        const long long work = (long long)P.nrows * ((P.W + 3) / 4) * planes;
        int pp = 8, g = 2;
        while (pp > 1 && work / pp < 131072) pp >>= 1;
        if (pp < 8) g = 4;
        if (const char* e = probe_env("MDTILE_BLEND_CFG")) {  // probes build only; "0,0" keeps the heuristic
            int epp = 0, eg = 0;
            if (sscanf(e, "%d,%d", &epp, &eg) == 2 && epp > 0 && eg > 0) { pp = epp; g = eg; }
        }
        while (pp > 1 && planes % pp != 0) pp >>= 1;
        if (pp >= 8 && g >= 4) launch_blend_cfg<T, 8, 4>(P, method, s);
        else if (pp >= 8) launch_blend_cfg<T, 8, 2>(P, method, s);
        else if (pp >= 4 && g >= 4) launch_blend_cfg<T, 4, 4>(P, method, s);
        else if (pp >= 4) launch_blend_cfg<T, 4, 2>(P, method, s);
        else if (pp >= 2) launch_blend_cfg<T, 2, 4>(P, method, s);
        else launch_blend_cfg<T, 1, 4>(P, method, s);
    }
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

int fill_params(BlendParams& P, const mdtile_plan* p, const mdtile_blend_args* a, const void* const* batch_out, int num_batches,
                const mdtile_region* regions, int num_regions) {
    MDT_CHECK_ARG(p && a, "mdtile_blend: null plan/args");
    if (int rc = plan_upload(p)) return rc;
    MDT_CHECK_ARG(a->N > 0 && a->C > 0 && a->N * a->C <= 65535, "mdtile_blend: bad N=%d C=%d", a->N, a->C);
    MDT_CHECK_ARG(a->method == MDTILE_METHOD_MD || a->method == MDTILE_METHOD_MOD, "mdtile_blend: bad method %d", a->method);
    MDT_CHECK_ARG(a->dtype >= 0 && a->dtype <= 2, "mdtile_blend: bad dtype %d", a->dtype);
    MDT_CHECK_ARG(num_regions >= 0 && (num_regions == 0 || regions), "mdtile_blend: bad regions");
    if (num_regions > MDTILE_MAX_REGIONS) {
        set_error("mdtile_blend: %d regions > MDTILE_MAX_REGIONS=%d", num_regions, MDTILE_MAX_REGIONS);
        return MDTILE_E_LIMIT;
    }
    MDT_CHECK_ARG(a->d_x_out, "mdtile_blend: null output");
    memset(&P, 0, sizeof(P));
    P.W = p->w; P.H = p->h; P.tw = p->tw; P.th = p->th; P.cols = p->cols; P.tile_bs = p->tile_bs; P.N = a->N; P.C = a->C;
    P.flags = a->flags; P.tile_lo = a->tile_lo; P.tile_hi = a->tile_hi;
    if (a->row_lo == 0 && a->row_hi == 0) { P.row_lo = 0; P.nrows = p->h; }
    else {
        MDT_CHECK_ARG(a->row_lo >= 0 && a->row_hi > a->row_lo && a->row_hi <= p->h, "mdtile_blend: bad row range [%d,%d)", a->row_lo, a->row_hi);
        P.row_lo = a->row_lo; P.nrows = a->row_hi - a->row_lo;
    }
    P.num_regions = num_regions; P.num_batches = num_batches;
    P.xs = p->d_xs; P.ys = p->d_ys; P.colrange = p->d_colrange; P.rowrange = p->d_rowrange;
    P.colquad = p->d_colquad; P.rowinfo = p->d_rowinfo;
    P.weights = a->d_weights; P.tile_w = a->d_tile_w; P.rescale = a->d_rescale; P.out = a->d_x_out;
    if (a->method == MDTILE_METHOD_MD && !(a->flags & MDTILE_BLEND_PARTIAL))
        MDT_CHECK_ARG(a->d_weights, "mdtile_blend: MultiDiffusion needs d_weights");
    if (a->method == MDTILE_METHOD_MOD && num_batches > 0)
        MDT_CHECK_ARG(a->d_tile_w && a->d_rescale, "mdtile_blend: Mixture of Diffusers needs d_tile_w and d_rescale");
    for (int k = 0; k < num_regions; ++k) {
        const mdtile_region& R = regions[k];
        MDT_CHECK_ARG(R.x >= 0 && R.y >= 0 && R.w > 0 && R.h > 0 && R.x + R.w <= p->w && R.y + R.h <= p->h && R.out,
                      "mdtile_blend: region %d (%d,%d,%d,%d) invalid", k, R.x, R.y, R.w, R.h);
        MDT_CHECK_ARG(R.mode == MDTILE_REGION_BG || R.mode == MDTILE_REGION_FG, "mdtile_blend: region %d bad mode", k);
        if (R.mode == MDTILE_REGION_FG || a->method == MDTILE_METHOD_MOD)
            MDT_CHECK_ARG(R.weight, "mdtile_blend: region %d needs a weight/feather map", k);
        P.regions[k] = R;
        P.num_fg += R.mode == MDTILE_REGION_FG;
    }
    if (num_batches > 0) {
        MDT_CHECK_ARG(batch_out, "mdtile_blend: null batch_out");
        if (a->flags & MDTILE_BLEND_PACKED) {
            MDT_CHECK_ARG(batch_out[0], "mdtile_blend: null packed buffer");
            P.batch[0] = batch_out[0];
        } else {
            MDT_CHECK_ARG(num_batches == p->num_batches, "mdtile_blend: %d batches given, plan has %d", num_batches, p->num_batches);
            if (num_batches > MDTILE_MAX_BATCHES) {
                set_error("mdtile_blend: %d batches > MDTILE_MAX_BATCHES=%d (use MDTILE_BLEND_PACKED)", num_batches, MDTILE_MAX_BATCHES);
                return MDTILE_E_LIMIT;
            }
            for (int b = 0; b < num_batches; ++b) {
                MDT_CHECK_ARG(batch_out[b], "mdtile_blend: null batch pointer %d", b);
                P.batch[b] = batch_out[b];
            }
        }
    }
    return MDTILE_OK;
}

}  // namespace

extern "C" int mdtile_blend(const mdtile_plan* plan, const mdtile_blend_args* args, const void* const* batch_out, int num_batches,
                            const mdtile_region* regions, int num_regions, mdtile_stream_t stream) {
    BlendParams P;
    int rc = fill_params(P, plan, args, batch_out, num_batches, regions, num_regions);
    if (rc != MDTILE_OK) return rc;
    hipStream_t s = as_stream(stream);
    switch (args->dtype) {
        case MDTILE_DT_F32: return launch_blend<float>(P, plan, args->method, false, s);
        case MDTILE_DT_F16: return launch_blend<__half>(P, plan, args->method, false, s);
        default: return launch_blend<__hip_bfloat16>(P, plan, args->method, false, s);
    }
}

extern "C" int mdtile_blend_finalize(const mdtile_plan* plan, const mdtile_blend_args* args, const float* d_partial,
                                     const mdtile_region* regions, int num_regions, mdtile_stream_t stream) {
    BlendParams P;
    MDT_CHECK_ARG(d_partial, "mdtile_blend_finalize: null partial buffer");
    mdtile_blend_args a = *args;
    a.flags &= ~MDTILE_BLEND_PARTIAL;
    int rc = fill_params(P, plan, &a, nullptr, 0, regions, num_regions);
    if (rc != MDTILE_OK) return rc;
    P.partial = d_partial;
    hipStream_t s = as_stream(stream);
    switch (args->dtype) {
        case MDTILE_DT_F32: return launch_blend<float>(P, nullptr, args->method, true, s);
        case MDTILE_DT_F16: return launch_blend<__half>(P, nullptr, args->method, true, s);
        default: return launch_blend<__hip_bfloat16>(P, nullptr, args->method, true, s);
    }
}

static int gather_common(const mdtile_plan* p, int dtype, int N, int C, const void* d_x_in, void* const* ptrs, int nptrs,
                         int t_lo, int t_hi, int packed, hipStream_t s) {
    MDT_CHECK_ARG(p && d_x_in && ptrs, "mdtile_gather: null argument");
    if (int rc = plan_upload(p)) return rc;
    MDT_CHECK_ARG(N > 0 && C > 0 && N * C <= 65535, "mdtile_gather: bad N=%d C=%d", N, C);
    MDT_CHECK_ARG(dtype >= 0 && dtype <= 2, "mdtile_gather: bad dtype %d", dtype);
    MDT_CHECK_ARG(t_hi - t_lo <= 65535, "mdtile_gather: too many tiles in one launch");
    if (nptrs > MDTILE_MAX_BATCHES) {
        set_error("mdtile_gather: %d batches > MDTILE_MAX_BATCHES=%d", nptrs, MDTILE_MAX_BATCHES);
        return MDTILE_E_LIMIT;
    }
    GatherParams P;
    memset(&P, 0, sizeof(P));
    P.W = p->w; P.H = p->h; P.tw = p->tw; P.th = p->th; P.cols = p->cols; P.tile_bs = p->tile_bs; P.N = N; P.C = C;
    P.t_lo = t_lo; P.t_hi = t_hi; P.packed = packed; P.xs = p->d_xs; P.ys = p->d_ys; P.x_in = d_x_in;
    for (int b = 0; b < nptrs; ++b) P.batch[b] = ptrs[b];  // holes are fine: only tiles in [t_lo, t_hi) are touched
    dim3 grid(cdiv((long long)p->th * ((p->tw + 3) / 4), 256), N * C, t_hi - t_lo), block(256);
    switch (dtype) {
        case MDTILE_DT_F32: hipLaunchKernelGGL(k_gather<float>, grid, block, 0, s, P); break;
        case MDTILE_DT_F16: hipLaunchKernelGGL(k_gather<__half>, grid, block, 0, s, P); break;
        default: hipLaunchKernelGGL(k_gather<__hip_bfloat16>, grid, block, 0, s, P); break;
    }
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_gather(const mdtile_plan* p, int dtype, int N, int C, const void* d_x_in, int batch_id, void* d_x_tile,
                             mdtile_stream_t stream) {
    MDT_CHECK_ARG(p && batch_id >= 0 && batch_id < p->num_batches && d_x_tile, "mdtile_gather: bad batch_id %d", batch_id);
    if (p->num_batches > MDTILE_MAX_BATCHES) {
        set_error("mdtile_gather: plan has %d batches > MDTILE_MAX_BATCHES=%d", p->num_batches, MDTILE_MAX_BATCHES);
        return MDTILE_E_LIMIT;
    }
    void* ptrs[MDTILE_MAX_BATCHES] = {nullptr};
    ptrs[batch_id] = d_x_tile;
    int t_lo = batch_id * p->tile_bs;
    int t_hi = t_lo + p->tile_bs < p->T ? t_lo + p->tile_bs : p->T;
    return gather_common(p, dtype, N, C, d_x_in, ptrs, p->num_batches, t_lo, t_hi, 0, as_stream(stream));
}

extern "C" int mdtile_gather_all(const mdtile_plan* p, int dtype, int N, int C, const void* d_x_in, void* const* batch_ptrs,
                                 int num_batches, mdtile_stream_t stream) {
    MDT_CHECK_ARG(p && batch_ptrs, "mdtile_gather_all: null argument");
    // num_batches == 1 with a plan of more batches means ONE packed [T*N,C,th,tw] destination
    int packed = (num_batches == 1 && p->num_batches != 1) ? 1 : 0;
    MDT_CHECK_ARG(packed || num_batches == p->num_batches, "mdtile_gather_all: %d batches given, plan has %d", num_batches, p->num_batches);
    for (int b = 0; b < num_batches; ++b) MDT_CHECK_ARG(batch_ptrs[b], "mdtile_gather_all: null batch pointer %d", b);
    return gather_common(p, dtype, N, C, d_x_in, batch_ptrs, num_batches, 0, p->T, packed, as_stream(stream));
}

extern "C" int mdtile_gather_range(const mdtile_plan* p, int dtype, int N, int C, const void* d_x_in, void* d_packed, int tile_lo,
                                   int tile_hi, mdtile_stream_t stream) {
    MDT_CHECK_ARG(p && d_packed, "mdtile_gather_range: null argument");
    MDT_CHECK_ARG(tile_lo >= 0 && tile_hi <= p->T && tile_lo <= tile_hi, "mdtile_gather_range: bad tile range [%d,%d) of %d", tile_lo, tile_hi, p->T);
    if (tile_lo == tile_hi) return MDTILE_OK;
    void* ptrs[1] = {d_packed};
    return gather_common(p, dtype, N, C, d_x_in, ptrs, 1, tile_lo, tile_hi, 1, as_stream(stream));
}

extern "C" int mdtile_stream_copy(const void* d_src, void* d_dst, size_t bytes, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_src && d_dst, "mdtile_stream_copy: null argument");
    MDT_CHECK_ARG(bytes % 16 == 0 && ((uintptr_t)d_src & 15) == 0 && ((uintptr_t)d_dst & 15) == 0, "mdtile_stream_copy: pointers and size must be multiples of 16 bytes");
    if (bytes == 0) return MDTILE_OK;
    const size_t n16 = bytes / 16;
    MDT_CHECK_ARG(n16 <= (size_t)0x7fffffff * 1024, "mdtile_stream_copy: %zu bytes in one launch", bytes);
    hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, as_stream(stream), (const u32x4v*)d_src, (u32x4v*)d_dst, n16);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_gather_rect(int dtype, int N, int C, int W, int H, const void* d_x_in, int x, int y, int w, int h,
                                  void* d_out, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x_in && d_out && N > 0 && C > 0 && N * C <= 65535, "mdtile_gather_rect: bad arguments");
    MDT_CHECK_ARG(x >= 0 && y >= 0 && w > 0 && h > 0 && x + w <= W && y + h <= H, "mdtile_gather_rect: rect (%d,%d,%d,%d) outside %dx%d", x, y, w, h, W, H);
    MDT_CHECK_ARG(dtype >= 0 && dtype <= 2, "mdtile_gather_rect: bad dtype %d", dtype);
    dim3 grid(cdiv((long long)w * h, 256), N * C), block(256);
    hipStream_t s = as_stream(stream);
    switch (dtype) {
        case MDTILE_DT_F32:
            hipLaunchKernelGGL(k_gather_rect<float>, grid, block, 0, s, (const float*)d_x_in, (float*)d_out, C, W, H, x, y, w, h);
            break;
        case MDTILE_DT_F16:
            hipLaunchKernelGGL(k_gather_rect<__half>, grid, block, 0, s, (const __half*)d_x_in, (__half*)d_out, C, W, H, x, y, w, h);
            break;
        default:
            hipLaunchKernelGGL(k_gather_rect<__hip_bfloat16>, grid, block, 0, s, (const __hip_bfloat16*)d_x_in, (__hip_bfloat16*)d_out, C, W, H, x, y, w, h);
            break;
    }
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}
