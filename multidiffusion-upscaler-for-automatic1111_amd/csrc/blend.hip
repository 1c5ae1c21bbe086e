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
#include "common.h"

using namespace mdt;

namespace {

struct BlendParams {
    int W, H, tw, th, cols, tile_bs, N, C;
    int flags, tile_lo, tile_hi, row_lo, nrows, num_regions, num_batches, _pad;
    const int *xs, *ys, *colrange, *rowrange;
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

// One thread owns 4 consecutive canvas columns of one row for PP (n,c) planes at once: the covering-tile lookup
// (rowrange / colrange / xs / ys, four small dependent loads) is done once and amortised over PP planes, and the PP
// 16-byte tile loads of one covering tile are independent -> PP loads in flight per lane before the first use.
// Plane p = n*C + c sits at `tile base + p * th*tw` in the tile-major batch layout, so the planes of one tile are a
// fixed stride apart.
template <typename T>
__device__ __forceinline__ const T* tile_base(const BlendParams& P, int t) {
    const size_t tile_elems = (size_t)P.th * P.tw;
    if (P.flags & MDTILE_BLEND_PACKED) return reinterpret_cast<const T*>(P.batch[0]) + (size_t)t * P.N * P.C * tile_elems;
    const int b = t / P.tile_bs, i = t - b * P.tile_bs;
    return reinterpret_cast<const T*>(P.batch[b]) + (size_t)i * P.N * P.C * tile_elems;
}

template <typename T, int METHOD, int PP>
__global__ __launch_bounds__(256) void k_blend(const BlendParams P) {
    const int W4 = (P.W + 3) >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W4 * P.nrows) return;
    const int yq = idx / W4;
    const int y = P.row_lo + yq;
    const int x0 = (idx - yq * W4) << 2;
    const int planes = P.N * P.C;
    const int p0 = blockIdx.y * PP;
    const int np = planes - p0 < PP ? planes - p0 : PP;   // wave-uniform
    const int nvalid = P.W - x0 < 4 ? P.W - x0 : 4;
    const size_t tile_elems = (size_t)P.th * P.tw;

    float acc[PP][4];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[pp][j] = 0.f;

    if (P.num_batches > 0) {
        const int rr = P.rowrange[y];
        const int r0 = rr & 0xffff, nr = rr >> 16;
        const int cr0 = P.colrange[x0];
        const bool uniform = nvalid == 4 && cr0 == P.colrange[x0 + 3];  // ranges are monotone: ends equal => all equal
        if (uniform) {
            const int c0 = cr0 & 0xffff, nc = cr0 >> 16;
            float resc[4];
            if (METHOD == MDTILE_METHOD_MOD) load4<float>(P.rescale + (size_t)y * P.W + x0, resc);
            for (int r = r0; r < r0 + nr; ++r) {
                const int ty = y - P.ys[r];
                for (int cc = c0; cc < c0 + nc; ++cc) {
                    const int t = r * P.cols + cc;
                    if ((P.flags & MDTILE_BLEND_TILE_RANGE) && (t < P.tile_lo || t >= P.tile_hi)) continue;
                    const size_t toff = (size_t)ty * P.tw + (x0 - P.xs[cc]);
                    const T* src = tile_base<T>(P, t) + (size_t)p0 * tile_elems + toff;
                    float v[PP][4];
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp)
                        if (pp < np) load4<T>(src + (size_t)pp * tile_elems, v[pp]);
                    if (METHOD == MDTILE_METHOD_MOD) {
                        float g[4], wgt[4];
                        load4<float>(P.tile_w + toff, g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) wgt[j] = g[j] * resc[j];  // w = tile_weights * rescale_factor[slicer]  (mixtureofdiffusers.py:125)
#pragma unroll
                        for (int pp = 0; pp < PP; ++pp)
                            if (pp < np)
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[pp][j] += v[pp][j] * wgt[j];  // x_buffer[slicer] += out * w  (:126)
                    } else {
#pragma unroll
                        for (int pp = 0; pp < PP; ++pp)
                            if (pp < np)
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[pp][j] += v[pp][j];           // multidiffusion.py:167
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= nvalid) continue;
                const int x = x0 + j;
                const int cr = P.colrange[x];
                const int c0 = cr & 0xffff, nc = cr >> 16;
                for (int r = r0; r < r0 + nr; ++r) {
                    const int ty = y - P.ys[r];
                    for (int cc = c0; cc < c0 + nc; ++cc) {
                        const int t = r * P.cols + cc;
                        if ((P.flags & MDTILE_BLEND_TILE_RANGE) && (t < P.tile_lo || t >= P.tile_hi)) continue;
                        const size_t toff = (size_t)ty * P.tw + (x - P.xs[cc]);
                        const T* src = tile_base<T>(P, t) + (size_t)p0 * tile_elems + toff;
                        float wgt = 1.0f;
                        if (METHOD == MDTILE_METHOD_MOD) wgt = P.tile_w[toff] * P.rescale[(size_t)y * P.W + x];
#pragma unroll
                        for (int pp = 0; pp < PP; ++pp) {
                            if (pp >= np) continue;
                            const float v = to_f32<T>(src[(size_t)pp * tile_elems]);
                            if (METHOD == MDTILE_METHOD_MOD) acc[pp][j] += v * wgt;
                            else acc[pp][j] += v;
                        }
                    }
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
                if (pp >= np) continue;
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
            if (pp >= np) continue;
            float* dst = reinterpret_cast<float*>(P.out) + o + (size_t)pp * plane_px;
            if (nvalid == 4) store4<float>(dst, acc[pp]);
            else for (int j = 0; j < nvalid; ++j) dst[j] = acc[pp][j];
        }
        return;
    }

    // MD normalisation: x = where(weights > 1, buf / weights, buf)  (multidiffusion.py:208); the weight is per pixel, shared by planes
    if (METHOD == MDTILE_METHOD_MD) {
        float w[4] = {1.f, 1.f, 1.f, 1.f};
        if (nvalid == 4) load4<float>(P.weights + (size_t)y * P.W + x0, w);
        else for (int j = 0; j < nvalid; ++j) w[j] = P.weights[(size_t)y * P.W + x0 + j];
#pragma unroll
        for (int pp = 0; pp < PP; ++pp)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[pp][j] = w[j] > 1.0f ? acc[pp][j] / w[j] : acc[pp][j];
    }

    // foreground feather composite (multidiffusion.py:191-198, 211-216 == mixtureofdiffusers.py:154-161, 170-175)
    bool any_fg = false;
    for (int k = 0; k < P.num_regions; ++k) any_fg |= P.regions[k].mode == MDTILE_REGION_FG;
    if (any_fg) {
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
                for (int pp = 0; pp < PP; ++pp)
                    if (pp < np) fbuf[pp] += to_f32<T>(src[(size_t)pp * rplane]);
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
        if (pp >= np) continue;
        T* dst = reinterpret_cast<T*>(P.out) + o + (size_t)pp * plane_px;
        if (nvalid == 4) store4<T>(dst, acc[pp]);
        else for (int j = 0; j < nvalid; ++j) dst[j] = from_f32<T>(acc[pp][j]);
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

template <typename T>
int launch_blend(const BlendParams& P, int method, bool finalize, hipStream_t s) {
    dim3 block(256);
    if (finalize) {
        dim3 grid(cdiv((long long)P.nrows * P.W, 256), P.N * P.C);
        if (method == MDTILE_METHOD_MD) hipLaunchKernelGGL((k_blend_finalize<T, MDTILE_METHOD_MD>), grid, block, 0, s, P);
        else hipLaunchKernelGGL((k_blend_finalize<T, MDTILE_METHOD_MOD>), grid, block, 0, s, P);
    } else {
        constexpr int PP = 8;  // planes per thread: N*C = 8 for batch-1 CFG (cond + uncond) x 4 latent channels
        dim3 grid(cdiv((long long)P.nrows * ((P.W + 3) / 4), 256), cdiv(P.N * P.C, PP));
        if (method == MDTILE_METHOD_MD) hipLaunchKernelGGL((k_blend<T, MDTILE_METHOD_MD, PP>), grid, block, 0, s, P);
        else hipLaunchKernelGGL((k_blend<T, MDTILE_METHOD_MOD, PP>), grid, block, 0, s, P);
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
