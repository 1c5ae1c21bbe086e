// Init-time weight maps (K8-K10).  Tiny, launch-latency-bound kernels; written for exact parity, not speed.
// Upstream: tile_utils/utils.py:160-214, tile_methods/multidiffusion.py:44-46, tile_methods/mixtureofdiffusers.py:29-55.
#include "common.h"

using namespace mdt;

// gaussian_weights (utils.py:187-194).  The profile is evaluated in fp64 exactly in upstream's operation order and
// only the final outer product is rounded to fp32.
__global__ void k_gaussian(int tw, int th, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tw * th) return;
    int y = i / tw, x = i - y * tw;
    const double var = 0.01;
    const double norm = sqrt(2.0 * 3.141592653589793 * var);
    const double tw2 = (double)(tw * tw);
    double dx = (double)x - (double)(tw - 1) / 2.0;
    double dy = (double)y - (double)th / 2.0;
    double px = exp(-dx * dx / tw2 / (2.0 * var)) / norm;
    double py = exp(-dy * dy / tw2 / (2.0 * var)) / norm;
    out[i] = (float)(py * px);
}

// feather_mask (utils.py:196-214)
__global__ void k_feather(int w, int h, int radius, float* __restrict__ out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w * h) return;
    int y = idx / w, x = idx - y * w;
    int i = y < h - 1 - y ? y : h - 1 - y;
    int j = x < w - 1 - x ? x : w - 1 - x;
    float v = 1.0f;
    if (i < h / 2 && j < w / 2) {
        int d = i < j ? i : j;
        if (d < radius) {
            double q = (double)d / (double)radius;
            v = (float)(q * q);
        }
    }
    out[idx] = v;
}

// weights[p] += sum over covering grid tiles (in upstream list order) of tile_w[...] (or 1.0)
__global__ void k_weight_grid(int W, int H, int tw, int cols, const int* __restrict__ xs, const int* __restrict__ ys,
                              const int* __restrict__ colrange, const int* __restrict__ rowrange,
                              const float* __restrict__ tile_w, float* __restrict__ weights) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= W * H) return;
    int y = idx / W, x = idx - y * W;
    int cr = colrange[x], rr = rowrange[y];
    int c0 = cr & 0xffff, nc = cr >> 16, r0 = rr & 0xffff, nr = rr >> 16;
    float s = 0.0f;
    for (int r = r0; r < r0 + nr; ++r) {
        int ty = y - ys[r];
        for (int c = c0; c < c0 + nc; ++c) {
            int tx = x - xs[c];
            s += tile_w ? tile_w[ty * tw + tx] : 1.0f;
        }
    }
    weights[idx] += s;
}

__global__ void k_weight_rect(float* __restrict__ weights, int W, int x0, int y0, int w, int h,
                              const float* __restrict__ rect_w, float scalar) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w * h) return;
    int y = idx / w, x = idx - y * w;
    weights[(size_t)(y0 + y) * W + x0 + x] += rect_w ? rect_w[idx] : scalar;
}

__global__ void k_recip(const float* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 1.0f / in[i];  // correctly rounded (hipcc default); inf where in == 0, as upstream
}

__global__ void k_rect_mul(float* __restrict__ rect_w, const float* __restrict__ canvas, int W, int x0, int y0, int w, int h) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w * h) return;
    int y = idx / w, x = idx - y * w;
    rect_w[idx] *= canvas[(size_t)(y0 + y) * W + x0 + x];
}

extern "C" int mdtile_gaussian_weights(int tile_w, int tile_h, float* d_out, mdtile_stream_t stream) {
    MDT_CHECK_ARG(tile_w > 0 && tile_h > 0 && d_out, "mdtile_gaussian_weights: bad arguments");
    int n = tile_w * tile_h;
    hipLaunchKernelGGL(k_gaussian, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), tile_w, tile_h, d_out);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_feather_mask(int w, int h, double ratio, float* d_out, mdtile_stream_t stream) {
    MDT_CHECK_ARG(w > 0 && h > 0 && d_out, "mdtile_feather_mask: bad arguments");
    int half = (w / 2) < (h / 2) ? (w / 2) : (h / 2);
    // Python: int(min(w//2, h//2) * ratio) with `ratio` a Python float, i.e. a double -- hence the double in the ABI
    int radius = (int)((double)half * ratio);
    int n = w * h;
    hipLaunchKernelGGL(k_feather, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), w, h, radius, d_out);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_weight_map_add_grid(const mdtile_plan* p, const float* d_tile_w, float* d_weights, mdtile_stream_t stream) {
    MDT_CHECK_ARG(p && d_weights, "mdtile_weight_map_add_grid: null argument");
    if (int rc = plan_upload(p)) return rc;
    int n = p->w * p->h;
    hipLaunchKernelGGL(k_weight_grid, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), p->w, p->h, p->tw, p->cols,
                       p->d_xs, p->d_ys, p->d_colrange, p->d_rowrange, d_tile_w, d_weights);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_weight_map_add_rect(float* d_weights, int W, int H, int x, int y, int w, int h, const float* d_rect_w,
                                          float scalar, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_weights && x >= 0 && y >= 0 && w > 0 && h > 0 && x + w <= W && y + h <= H,
                  "mdtile_weight_map_add_rect: rect (%d,%d,%d,%d) outside %dx%d", x, y, w, h, W, H);
    int n = w * h;
    hipLaunchKernelGGL(k_weight_rect, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), d_weights, W, x, y, w, h, d_rect_w, scalar);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_reciprocal(const float* d_in, float* d_out, size_t n, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_in && d_out, "mdtile_reciprocal: null argument");
    if (n == 0) return MDTILE_OK;
    hipLaunchKernelGGL(k_recip, dim3(cdiv((long long)n, 256)), dim3(256), 0, as_stream(stream), d_in, d_out, n);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_rect_mul_canvas(float* d_rect_w, const float* d_canvas, int W, int H, int x, int y, int w, int h,
                                      mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_rect_w && d_canvas && x >= 0 && y >= 0 && w > 0 && h > 0 && x + w <= W && y + h <= H,
                  "mdtile_rect_mul_canvas: rect (%d,%d,%d,%d) outside %dx%d", x, y, w, h, W, H);
    int n = w * h;
    hipLaunchKernelGGL(k_rect_mul, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), d_rect_w, d_canvas, W, x, y, w, h);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}
