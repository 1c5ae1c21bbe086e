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

// ---------------------------------------------------------------------------------------------------------------------
// Per-region initial noise (scripts/tilediffusion.py:486-529, create_random_tensors_hijack): every enabled region draws its
// own CPU-seeded noise [1, C, h, w]; background and foreground regions are summed into two layers with hit counts, each layer
// is averaged where more than one region covers a pixel, and pasted over the job's noise -- background first, foreground on
// top.  Gather-formulated like the blend: one thread owns one element of the output, walks the regions in upstream's order
// (fp32 sums in the same order as the sequential `+=`, counts are exact small integers) -> bit-identical to upstream.
namespace {
constexpr int MAX_NOISE_REGIONS = 16;   // upstream caps the UI at 16 regions (BBOX_MAX_NUM)
struct NoiseRegions {
    int x[MAX_NOISE_REGIONS], y[MAX_NOISE_REGIONS], w[MAX_NOISE_REGIONS], h[MAX_NOISE_REGIONS], mode[MAX_NOISE_REGIONS];
    const float* rand[MAX_NOISE_REGIONS];
    int n;
};

__global__ __launch_bounds__(256) void k_region_noise(float* __restrict__ noise, int C, int H, int W, const NoiseRegions R) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= H * W) return;
    const int plane = blockIdx.y, c = plane % C;     // plane = n*C + c; the region noise is shared by every sample n
    const int y = px / W, x = px - y * W;
    float bg = 0.0f, bgc = 0.0f, fg = 0.0f, fgc = 0.0f;
    for (int r = 0; r < R.n; ++r) {
        const int dx = x - R.x[r], dy = y - R.y[r];
        if (dx < 0 || dy < 0 || dx >= R.w[r] || dy >= R.h[r]) continue;
        const float v = R.rand[r][((size_t)c * R.h[r] + dy) * R.w[r] + dx];
        if (R.mode[r] == MDTILE_REGION_BG) { bg += v; bgc += 1.0f; }
        else { fg += v; fgc += 1.0f; }
    }
    const size_t o = (size_t)plane * H * W + px;
    float out = noise[o];
    if (bgc > 0.0f) out = bgc > 1.0f ? bg / bgc : bg;
    if (fgc > 0.0f) out = fgc > 1.0f ? fg / fgc : fg;
    noise[o] = out;
}
}  // namespace

extern "C" int mdtile_region_noise(float* d_noise, int N, int C, int H, int W, const mdtile_region* regions, int num_regions,
                                   mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_noise && N > 0 && C > 0 && H > 0 && W > 0 && N * C <= 65535, "mdtile_region_noise: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
    MDT_CHECK_ARG(num_regions >= 0 && num_regions <= MAX_NOISE_REGIONS && (num_regions == 0 || regions), "mdtile_region_noise: %d regions (max %d)",
                  num_regions, MAX_NOISE_REGIONS);
    if (num_regions == 0) return MDTILE_OK;
    NoiseRegions R;
    R.n = num_regions;
    for (int i = 0; i < num_regions; ++i) {
        const mdtile_region& g = regions[i];
        MDT_CHECK_ARG(g.out && g.w > 0 && g.h > 0 && g.x >= 0 && g.y >= 0 && g.x + g.w <= W && g.y + g.h <= H,
                      "mdtile_region_noise: region %d rect (%d,%d,%d,%d) outside %dx%d or without noise", i, g.x, g.y, g.w, g.h, W, H);
        MDT_CHECK_ARG(g.mode == MDTILE_REGION_BG || g.mode == MDTILE_REGION_FG, "mdtile_region_noise: region %d has mode %d", i, g.mode);
        R.x[i] = g.x; R.y[i] = g.y; R.w[i] = g.w; R.h[i] = g.h; R.mode[i] = g.mode; R.rand[i] = (const float*)g.out;
    }
    hipLaunchKernelGGL(k_region_noise, dim3(cdiv((long long)H * W, 256), N * C), dim3(256), 0, as_stream(stream), d_noise, C, H, W, R);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Noise Inversion: the renoise composite of sample_img2img (tile_methods/abstractdiffusion.py:651-676).
//   regions (only when the grid is disabled, :658-672):  background hit count; foreground: sum of the noise itself, of the feather
//   masks and the hit count, in list order;  bgn = bc > 0 ? noise : 0;  fgn = fc > 0 ? fsum / fc : 0;  fgw = fc > 0 ? wsum / fc : 0;
//   noise' = bgn * (1 - fgw) + fgn * fgw
//   combined = ((1 - m) * inverse + m * noise') / sqrt(m*m + (1 - m)*(1 - m))          m = renoise mask [H, W]
// One thread per element, the same fp32 operations in the same order as the eager code (this file is built with
// -ffp-contract=off; / and sqrt are correctly rounded -- torch's own CPU sqrt is not, the one possible last-bit difference).
namespace {
struct InvRegions {
    int x[MAX_NOISE_REGIONS], y[MAX_NOISE_REGIONS], w[MAX_NOISE_REGIONS], h[MAX_NOISE_REGIONS], mode[MAX_NOISE_REGIONS];
    const float* feather[MAX_NOISE_REGIONS];
    int n;
};

__global__ __launch_bounds__(256) void k_noise_inverse_blend(const float* __restrict__ noise, const float* __restrict__ inverse,
                                                             const float* __restrict__ mask, float* __restrict__ out, int H, int W,
                                                             const InvRegions R) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= H * W) return;
    const size_t o = (size_t)blockIdx.y * H * W + px;
    float n = noise[o];
    if (R.n > 0) {
        const int y = px / W, x = px - y * W;
        float bc = 0.0f, fsum = 0.0f, wsum = 0.0f, fc = 0.0f;
        for (int r = 0; r < R.n; ++r) {
            const int dx = x - R.x[r], dy = y - R.y[r];
            if (dx < 0 || dy < 0 || dx >= R.w[r] || dy >= R.h[r]) continue;
            if (R.mode[r] == MDTILE_REGION_BG) {
                bc += 1.0f;
            } else {
                fsum += n;
                wsum += R.feather[r][(size_t)dy * R.w[r] + dx];
                fc += 1.0f;
            }
        }
        const float bgn = bc > 0.0f ? n : 0.0f;
        const float fgn = fc > 0.0f ? fsum / fc : 0.0f;
        const float fgw = fc > 0.0f ? wsum / fc : 0.0f;
        n = bgn * (1.0f - fgw) + fgn * fgw;
    }
    const float m = mask[px], om = 1.0f - m;
    out[o] = (om * inverse[o] + m * n) / sqrtf(m * m + om * om);
}
}  // namespace

extern "C" int mdtile_noise_inverse_blend(const float* d_noise, const float* d_inverse_noise, const float* d_renoise_mask, float* d_out,
                                          int N, int C, int H, int W, const mdtile_region* regions, int num_regions,
                                          mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_noise && d_inverse_noise && d_renoise_mask && d_out, "mdtile_noise_inverse_blend: null argument");
    MDT_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && N * C <= 65535, "mdtile_noise_inverse_blend: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
    MDT_CHECK_ARG(num_regions >= 0 && num_regions <= MAX_NOISE_REGIONS && (num_regions == 0 || regions),
                  "mdtile_noise_inverse_blend: %d regions (max %d)", num_regions, MAX_NOISE_REGIONS);
    InvRegions R;
    R.n = num_regions;
    for (int i = 0; i < num_regions; ++i) {
        const mdtile_region& g = regions[i];
        MDT_CHECK_ARG(g.w > 0 && g.h > 0 && g.x >= 0 && g.y >= 0 && g.x + g.w <= W && g.y + g.h <= H,
                      "mdtile_noise_inverse_blend: region %d rect (%d,%d,%d,%d) outside %dx%d", i, g.x, g.y, g.w, g.h, W, H);
        MDT_CHECK_ARG(g.mode == MDTILE_REGION_BG || (g.mode == MDTILE_REGION_FG && g.weight),
                      "mdtile_noise_inverse_blend: region %d: mode %d (foreground regions need their feather mask)", i, g.mode);
        R.x[i] = g.x; R.y[i] = g.y; R.w[i] = g.w; R.h[i] = g.h; R.mode[i] = g.mode; R.feather[i] = g.weight;
    }
    hipLaunchKernelGGL(k_noise_inverse_blend, dim3(cdiv((long long)H * W, 256), N * C), dim3(256), 0, as_stream(stream), d_noise,
                       d_inverse_noise, d_renoise_mask, d_out, H, W, R);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ControlNet / StableSR tile slicing (tile_methods/abstractdiffusion.py:475-544, 548-588): up to 16 equally sized rectangles
// of a tensor [N, C, H, W] (the latent grid scaled by `opt_f` = 8 for ControlNet hints, by 1 for the StableSR latent image)
// are cut out, concatenated along the batch axis and repeated for the sampler's cond / uncond copies in ONE launch:
//     rows = cat_i x[:, :, rect_i]                     (i-major, then the N samples -- torch.cat over the bboxes)
//     tile_major = 1:  out row (j * repeat + r) = rows[j]      (k-diffusion: every tile's copies are consecutive, :528-533)
//     tile_major = 0:  out row (r * nrows + j) = rows[j]       (DDIM: the whole batch is repeated, :535)
// Pure copies: bit-exact against the slicing by construction; the per-batch CPU-side tile caches of upstream are not needed.
namespace {
constexpr int MAX_GATHER_RECTS = 16;
struct GatherRects {
    int x[MAX_GATHER_RECTS], y[MAX_GATHER_RECTS];
    int n;
};

template <typename T>
__global__ __launch_bounds__(256) void k_gather_rects(const T* __restrict__ x_in, T* __restrict__ out, int N, int C, int H, int W,
                                                      int w, int h, int repeat, int tile_major, const GatherRects R) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= w * h) return;
    const int y = idx / w, x = idx - y * w;
    const int c = blockIdx.y;
    const int orow = blockIdx.z, nrows = R.n * N;
    const int j = tile_major ? orow / repeat : orow % nrows;      // source row: tile i = j / N, sample n = j % N
    const int i = j / N, n = j - i * N;
    out[((size_t)orow * C + c) * w * h + idx] = x_in[(((size_t)n * C + c) * H + R.y[i] + y) * W + R.x[i] + x];
}
}  // namespace

extern "C" int mdtile_gather_rects(int dtype, int N, int C, int W, int H, const void* d_x_in, const int* rects_xy, int num_rects, int w, int h,
                                   int repeat, int tile_major, void* d_out, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x_in && d_out && rects_xy, "mdtile_gather_rects: null argument");
    MDT_CHECK_ARG(N > 0 && C > 0 && C <= 65535 && W > 0 && H > 0 && w > 0 && h > 0 && repeat > 0, "mdtile_gather_rects: bad shape");
    MDT_CHECK_ARG(num_rects > 0 && num_rects <= MAX_GATHER_RECTS, "mdtile_gather_rects: %d rects (1..%d)", num_rects, MAX_GATHER_RECTS);
    MDT_CHECK_ARG((long long)num_rects * N * repeat <= 65535, "mdtile_gather_rects: %d output rows", num_rects * N * repeat);
    GatherRects R;
    R.n = num_rects;
    for (int i = 0; i < num_rects; ++i) {
        R.x[i] = rects_xy[2 * i];
        R.y[i] = rects_xy[2 * i + 1];
        MDT_CHECK_ARG(R.x[i] >= 0 && R.y[i] >= 0 && R.x[i] + w <= W && R.y[i] + h <= H, "mdtile_gather_rects: rect %d (%d,%d,%d,%d) outside %dx%d", i,
                      R.x[i], R.y[i], w, h, W, H);
    }
    dim3 grid(cdiv((long long)w * h, 256), C, num_rects * N * repeat), block(256);
    hipStream_t s = as_stream(stream);
    switch (dtype) {
        case MDTILE_DT_F32:
            hipLaunchKernelGGL(k_gather_rects<float>, grid, block, 0, s, (const float*)d_x_in, (float*)d_out, N, C, H, W, w, h, repeat, tile_major, R);
            break;
        case MDTILE_DT_F16:
            hipLaunchKernelGGL(k_gather_rects<__half>, grid, block, 0, s, (const __half*)d_x_in, (__half*)d_out, N, C, H, W, w, h, repeat, tile_major, R);
            break;
        case MDTILE_DT_BF16:
            hipLaunchKernelGGL(k_gather_rects<__hip_bfloat16>, grid, block, 0, s, (const __hip_bfloat16*)d_x_in, (__hip_bfloat16*)d_out, N, C, H, W, w, h,
                               repeat, tile_major, R);
            break;
        default:
            MDT_CHECK_ARG(false, "mdtile_gather_rects: bad dtype %d", dtype);
    }
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}
