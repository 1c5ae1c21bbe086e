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
// Access pattern: a thread owns 4 consecutive canvas columns of one row for 8 (n,c) planes -> 16 B per lane and plane,
// 1 KiB contiguous per wave and plane on the store side; tile-side reads are 4-element vectors at an element-aligned
// (not 16 B-aligned) address because tile origins are arbitrary (gfx950 unaligned-access mode: one global_load_dwordx4).
// No LDS staging and no cross-lane reduction is needed in this formulation: the sum over covering tiles is a short
// in-register loop (1 tile for ~75 % of the pixels at overlap 8, at most 4 in the corners of the overlap lattice).
#include <type_traits>

#include "common.h"

using namespace mdt;


namespace {

struct BlendParams {
    int W, H, tw, th, cols, tile_bs, N, C;
    int flags, tile_lo, tile_hi, row_lo, nrows, num_regions, num_batches, num_fg;
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

template <typename T, bool PACKED>
__device__ __forceinline__ const T* tile_base(const BlendParams& P, int t) {
    const size_t tile_elems = (size_t)P.th * P.tw;
    if (PACKED) return reinterpret_cast<const T*>(P.batch[0]) + (size_t)t * P.N * P.C * tile_elems;
    const int b = t / P.tile_bs, i = t - b * P.tile_bs;   // P.batch[b] with a per-lane b is a (cached) vector load
    return reinterpret_cast<const T*>(P.batch[b]) + (size_t)i * P.N * P.C * tile_elems;
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
                            for (int pp = 0; pp < PP; ++pp) load4<T>(row[g] + (size_t)pp * tile_elems, v[g][pp]);
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
        if (nvalid == 4) store4<T>(dst, acc[pp]);
        else {
            dst[0] = from_f32<T>(acc[pp][0]);
            if (nvalid > 1) dst[1] = from_f32<T>(acc[pp][1]);
            if (nvalid > 2) dst[2] = from_f32<T>(acc[pp][2]);
        }
    }
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

// the blend's measurement floor (mdtile_stream_copy): every thread moves 4 x 16 bytes, a wave 4 x 1 KiB runs a block-stride apart
__global__ __launch_bounds__(256) void k_stream_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + i * 256 < n16) v[i] = src[base + i * 256];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + i * 256 < n16) dst[base + i * 256] = v[i];
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

template <typename T>
int launch_blend(const BlendParams& P, int method, bool finalize, hipStream_t s) {
    dim3 block(256);
    if (finalize) {
        dim3 grid(cdiv((long long)P.nrows * P.W, 256), P.N * P.C);
        if (method == MDTILE_METHOD_MD) hipLaunchKernelGGL((k_blend_finalize<T, MDTILE_METHOD_MD>), grid, block, 0, s, P);
        else hipLaunchKernelGGL((k_blend_finalize<T, MDTILE_METHOD_MOD>), grid, block, 0, s, P);
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
        case MDTILE_DT_F32: return launch_blend<float>(P, args->method, false, s);
        case MDTILE_DT_F16: return launch_blend<__half>(P, args->method, false, s);
        default: return launch_blend<__hip_bfloat16>(P, args->method, false, s);
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
        case MDTILE_DT_F32: return launch_blend<float>(P, args->method, true, s);
        case MDTILE_DT_F16: return launch_blend<__half>(P, args->method, true, s);
        default: return launch_blend<__hip_bfloat16>(P, args->method, true, s);
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
    hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, as_stream(stream), (const uint4*)d_src, (uint4*)d_dst, n16);
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
