// HBM-bound Tiled-VAE kernels: GroupNorm statistics (K11), pooling (K12), fixed-statistics GroupNorm + SiLU (K13),
// SiLU / residual add (K16), crop + assemble (K17), fast-mode estimator input (K18).
// Upstream: scripts/tilevae.py:102-104, 207-259, 289-361, 545-559, 612-632.  All tensors fp32 NCHW.
#include "common.h"

using namespace mdt;

namespace {

constexpr int GN_NBLK = 64;  // partial blocks per (sample, group)

__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
    // wave64 shuffle reduction on the two 32-bit halves, then one LDS hop across the block's waves
    for (int off = 32; off > 0; off >>= 1) {
        int lo = __shfl_xor(__double2loint(v), off, 64);
        int hi = __shfl_xor(__double2hiint(v), off, 64);
        v += __hiloint2double(hi, lo);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < nw; ++w) r += sh[w];
    return r;
}

// A (sample, group) is ONE contiguous run of L = (C/groups)*HW floats in NCHW.  Stage 1: GN_NBLK blocks per group
// stream their slice with 16-B loads and accumulate sum / sum-of-squares in fp64 (upstream's CPU var_mean also
// accumulates in double); stage 2 combines the partials in a fixed order -> deterministic, no atomics.
__global__ __launch_bounds__(256) void k_gn_partial(const float* __restrict__ x, size_t L, double* __restrict__ part) {
    __shared__ double sh[4];
    const int g = blockIdx.y, blk = blockIdx.x;
    const float* base = x + (size_t)g * L;
    double s1 = 0.0, s2 = 0.0;
    const bool vec_ok = ((reinterpret_cast<size_t>(base) & 15) == 0);
    if (vec_ok) {
        const size_t L4 = L >> 2;
        const float4* b4 = reinterpret_cast<const float4*>(base);
        for (size_t i = (size_t)blk * 256 + threadIdx.x; i < L4; i += (size_t)GN_NBLK * 256) {
            float4 v = b4[i];
            s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
        for (size_t i = (L4 << 2) + (size_t)blk * 256 + threadIdx.x; i < L; i += (size_t)GN_NBLK * 256) {
            double v = base[i];
            s1 += v; s2 += v * v;
        }
    } else {
        for (size_t i = (size_t)blk * 256 + threadIdx.x; i < L; i += (size_t)GN_NBLK * 256) {
            double v = base[i];
            s1 += v; s2 += v * v;
        }
    }
    s1 = block_sum_f64(s1, sh);
    s2 = block_sum_f64(s2, sh);
    if (threadIdx.x == 0) {
        part[((size_t)g * GN_NBLK + blk) * 2 + 0] = s1;
        part[((size_t)g * GN_NBLK + blk) * 2 + 1] = s2;
    }
}

__global__ void k_gn_final(const double* __restrict__ part, size_t L, int BG, float* __restrict__ mean, float* __restrict__ var) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= BG) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < GN_NBLK; ++b) {
        s1 += part[((size_t)g * GN_NBLK + b) * 2 + 0];
        s2 += part[((size_t)g * GN_NBLK + b) * 2 + 1];
    }
    double m = s1 / (double)L;
    double v = s2 / (double)L - m * m;  // biased (unbiased=False, tilevae.py:214)
    if (v < 0.0) v = 0.0;
    mean[g] = (float)m;
    var[g] = (float)v;
}

// Statistics left by a conv's epilogue (k_conv3x3_bf16x3<.., ST = true> and the record kernels' epilogue_item<.., ST>): fp64 partials
// part[b][unit][cb][QB quads][2] (unit = pixel tile of the producing kernel, cb = its cout block, quad = 4 consecutive couts).  Same
// two-stage shape as k_gn_partial -> k_gn_final: GN_NBLK blocks per (sample, group) add their units in a fixed order.
__global__ __launch_bounds__(256) void k_conv_stats_partial(const double* __restrict__ cpart, int units, int NCB, int QB, int cpg, int groups,
                                                            double* __restrict__ part) {
    __shared__ double sh[4];
    const int bg = blockIdx.y, blk = blockIdx.x, b = bg / groups, g = bg - b * groups;
    const int q_first = g * cpg / 4, nq = cpg / 4, cb = q_first / QB, q0 = q_first - cb * QB;
    double s1 = 0.0, s2 = 0.0;
    for (int u = blk * 256 + threadIdx.x; u < units; u += GN_NBLK * 256) {
        const double2* p = reinterpret_cast<const double2*>(cpart) + (((size_t)b * units + u) * NCB + cb) * QB + q0;
        for (int q = 0; q < nq; ++q) {
            const double2 v = p[q];
            s1 += v.x;
            s2 += v.y;
        }
    }
    s1 = block_sum_f64(s1, sh);
    s2 = block_sum_f64(s2, sh);
    if (threadIdx.x == 0) {
        part[((size_t)bg * GN_NBLK + blk) * 2 + 0] = s1;
        part[((size_t)bg * GN_NBLK + blk) * 2 + 1] = s2;
    }
}

// Row-band variant for the sequence-parallel estimator: the (sample, group) covers `cpg` planes of `plane_stride` floats, of
// which only [off, off + len) of every plane (this rank's own rows, halo rows excluded) enter the sums.  Output = the raw
// fp64 (sum, sum of squares) per group: ranks all-reduce them and finish with k_gn_from_sums.
__global__ __launch_bounds__(256) void k_gn_partial_rows(const float* __restrict__ x, int cpg, size_t plane_stride, size_t off, size_t len,
                                                         double* __restrict__ part) {
    __shared__ double sh[4];
    const int g = blockIdx.y, blk = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int p = 0; p < cpg; ++p) {
        const float* base = x + ((size_t)g * cpg + p) * plane_stride + off;
        for (size_t i = (size_t)blk * 256 + threadIdx.x; i < len; i += (size_t)GN_NBLK * 256) {
            const double v = base[i];
            s1 += v; s2 += v * v;
        }
    }
    s1 = block_sum_f64(s1, sh);
    s2 = block_sum_f64(s2, sh);
    if (threadIdx.x == 0) {
        part[((size_t)g * GN_NBLK + blk) * 2 + 0] = s1;
        part[((size_t)g * GN_NBLK + blk) * 2 + 1] = s2;
    }
}

__global__ void k_gn_sum_partials(const double* __restrict__ part, int BG, double* __restrict__ sums) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= BG) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < GN_NBLK; ++b) {
        s1 += part[((size_t)g * GN_NBLK + b) * 2 + 0];
        s2 += part[((size_t)g * GN_NBLK + b) * 2 + 1];
    }
    sums[(size_t)g * 2 + 0] = s1;
    sums[(size_t)g * 2 + 1] = s2;
}

__global__ void k_gn_from_sums(const double* __restrict__ sums, double count, int BG, float* __restrict__ mean, float* __restrict__ var) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= BG) return;
    const double m = sums[(size_t)g * 2] / count;
    double v = sums[(size_t)g * 2 + 1] / count - m * m;   // biased, as k_gn_final
    if (v < 0.0) v = 0.0;
    mean[g] = (float)m;
    var[g] = (float)v;
}

// tilevae.py:320-335: p_i = (px_i / max px) / sum(px_j / max px); var = sum p_i var_i; mean = sum p_i mean_i  (fp32)
__global__ void k_gn_pool(const float* __restrict__ means, const float* __restrict__ vars, const float* __restrict__ p, int T,
                          int BG, float* __restrict__ mean, float* __restrict__ var) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= BG) return;
    float m = 0.f, v = 0.f;
    for (int t = 0; t < T; ++t) {
        v += vars[(size_t)t * BG + g] * p[t];
        m += means[(size_t)t * BG + g] * p[t];
    }
    mean[g] = m;
    var[g] = v;
}

__device__ __forceinline__ float silu_f(float y) { return y / (1.0f + expf(-y)); }

// y = ((x - mean_g) * rstd_g) * gamma_c + beta_c  [-> SiLU]; one (b,c) plane per blockIdx.y; 16 B per lane.
template <bool SILU>
__global__ __launch_bounds__(256) void k_gn_apply(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int cpg,
                                                  int groups, const float* __restrict__ mean, const float* __restrict__ var,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps) {
    const int plane = blockIdx.y;  // b*C + c
    const int b = plane / C, c = plane - b * C;
    const int g = b * groups + c / cpg;
    const float rstd = 1.0f / sqrtf(var[g] + eps);
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    const float a = rstd * ga;
    const float sh = fmaf(-mean[g], a, be);
    const float* xp = x + (size_t)plane * HW;
    float* yp = y + (size_t)plane * HW;
    if ((((size_t)plane * HW) & 3) == 0 && ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(y)) & 15) == 0) {
        const int HW4 = HW >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW4; i += gridDim.x * 256) {
            float4 v = reinterpret_cast<const float4*>(xp)[i];
            float4 o;
            o.x = fmaf(v.x, a, sh); o.y = fmaf(v.y, a, sh); o.z = fmaf(v.z, a, sh); o.w = fmaf(v.w, a, sh);
            if (SILU) { o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w); }
            reinterpret_cast<float4*>(yp)[i] = o;
        }
        for (int i = (HW4 << 2) + blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
            float o = fmaf(xp[i], a, sh);
            yp[i] = SILU ? silu_f(o) : o;
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
            float o = fmaf(xp[i], a, sh);
            yp[i] = SILU ? silu_f(o) : o;
        }
    }
}

template <int OP> __device__ __forceinline__ float eltwise_f(float a, float b) {
    return OP == 0 ? silu_f(a) : OP == 1 ? a + b : tanhf(a);
}

template <int OP>  // 0 = silu(a), 1 = a + b, 2 = tanh(a)
__global__ __launch_bounds__(256) void k_eltwise(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, size_t n) {
    const size_t n4 = n >> 2;
    const bool vec = ((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(y) | (OP == 1 ? reinterpret_cast<size_t>(b) : 0)) & 15) == 0;
    const size_t stride = (size_t)gridDim.x * 256;
    if (vec) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            float4 v = reinterpret_cast<const float4*>(a)[i], o;
            if (OP != 1) { o.x = eltwise_f<OP>(v.x, 0.f); o.y = eltwise_f<OP>(v.y, 0.f); o.z = eltwise_f<OP>(v.z, 0.f); o.w = eltwise_f<OP>(v.w, 0.f); }
            else { float4 w = reinterpret_cast<const float4*>(b)[i]; o.x = v.x + w.x; o.y = v.y + w.y; o.z = v.z + w.z; o.w = v.w + w.w; }
            reinterpret_cast<float4*>(y)[i] = o;
        }
        for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = eltwise_f<OP>(a[i], OP == 1 ? b[i] : 0.f);
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = eltwise_f<OP>(a[i], OP == 1 ? b[i] : 0.f);
    }
}

// result[n,c, oy0+yy, ox0+xx] = tile[n,c, m2+yy, m0+xx]
__global__ __launch_bounds__(256) void k_crop_store(const float* __restrict__ tile, int th, int tw, int m0, int m2, int cw, int ch,
                                                    float* __restrict__ result, int RH, int RW, int ox0, int oy0) {
    const int plane = blockIdx.y;
    const size_t n = (size_t)cw * ch;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int yy = (int)(i / cw), xx = (int)(i - (size_t)yy * cw);
        result[((size_t)plane * RH + oy0 + yy) * RW + ox0 + xx] = tile[((size_t)plane * th + m2 + yy) * tw + m0 + xx];
    }
}

// ---- fast-mode estimator input (tilevae.py:545-559) ---------------------------------------------------------------
// ws layout (doubles): per channel c: [s1_z, s2_z, s1_d, s2_d]; then floats: min_z[c], max_z[c]
__device__ __forceinline__ int nearest_exact_src(int dst, float scale, int in_size) {
    int s = (int)floorf(((float)dst + 0.5f) * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

constexpr int FAST_NBLK = 128;   // partial blocks per channel (the encoder runs this on a whole 8K image: 67 M elements per channel)

// stage 1: block (blk, c) accumulates its grid-stride share of channel c: fp64 sums of z and of the down-sampled z, min / max
__global__ __launch_bounds__(256) void k_fast_stats(const float* __restrict__ z, int N, int C, int H, int W, int oh, int ow,
                                                    float sc_h, float sc_w, double* __restrict__ pd, float* __restrict__ pf) {
    __shared__ double sh[4];
    __shared__ float shf[8];
    const int c = blockIdx.y, blk = blockIdx.x;
    double s1 = 0, s2 = 0, d1 = 0, d2 = 0;
    float mn = INFINITY, mx = -INFINITY;
    const size_t stride = (size_t)FAST_NBLK * 256, start = (size_t)blk * 256 + threadIdx.x;
    for (int n = 0; n < N; ++n) {
        const float* p = z + ((size_t)n * C + c) * H * W;
        for (size_t i = start; i < (size_t)H * W; i += stride) {
            float v = p[i];
            s1 += v; s2 += (double)v * v;
            mn = fminf(mn, v); mx = fmaxf(mx, v);
        }
        for (size_t i = start; i < (size_t)oh * ow; i += stride) {
            int oy = (int)(i / ow), ox = (int)(i - (size_t)oy * ow);
            float v = p[(size_t)nearest_exact_src(oy, sc_h, H) * W + nearest_exact_src(ox, sc_w, W)];
            d1 += v; d2 += (double)v * v;
        }
    }
    s1 = block_sum_f64(s1, sh); s2 = block_sum_f64(s2, sh); d1 = block_sum_f64(d1, sh); d2 = block_sum_f64(d2, sh);
    mn = wave_min(mn); mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { shf[threadIdx.x >> 6] = mn; shf[4 + (threadIdx.x >> 6)] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = pd + ((size_t)c * FAST_NBLK + blk) * 4;
        o[0] = s1; o[1] = s2; o[2] = d1; o[3] = d2;
        pf[((size_t)c * FAST_NBLK + blk) * 2 + 0] = fminf(fminf(shf[0], shf[1]), fminf(shf[2], shf[3]));
        pf[((size_t)c * FAST_NBLK + blk) * 2 + 1] = fmaxf(fmaxf(shf[4], shf[5]), fmaxf(shf[6], shf[7]));
    }
}

// stage 2: one thread per channel combines the partials in a fixed order (deterministic)
__global__ void k_fast_combine(const double* __restrict__ pd, const float* __restrict__ pf, int C, double* __restrict__ wsd,
                               float* __restrict__ wsf) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a[4] = {0, 0, 0, 0};
    float mn = INFINITY, mx = -INFINITY;
    for (int b = 0; b < FAST_NBLK; ++b) {
        const double* o = pd + ((size_t)c * FAST_NBLK + b) * 4;
        a[0] += o[0]; a[1] += o[1]; a[2] += o[2]; a[3] += o[3];
        mn = fminf(mn, pf[((size_t)c * FAST_NBLK + b) * 2 + 0]);
        mx = fmaxf(mx, pf[((size_t)c * FAST_NBLK + b) * 2 + 1]);
    }
    wsd[c * 4 + 0] = a[0]; wsd[c * 4 + 1] = a[1]; wsd[c * 4 + 2] = a[2]; wsd[c * 4 + 3] = a[3];
    wsf[c] = mn;
    wsf[C + c] = mx;
}

__global__ __launch_bounds__(256) void k_fast_apply(const float* __restrict__ z, int N, int C, int H, int W, float* __restrict__ out,
                                                    int oh, int ow, float sc_h, float sc_w, const double* __restrict__ wsd,
                                                    const float* __restrict__ wsf) {
    const size_t total = (size_t)N * C * oh * ow;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
    const int c = (int)((i / ((size_t)ow * oh)) % C), n = (int)(i / ((size_t)ow * oh * C));
    float zmin = INFINITY, zmax = -INFINITY;
    for (int k = 0; k < C; ++k) { zmin = fminf(zmin, wsf[k]); zmax = fmaxf(zmax, wsf[C + k]); }
    const double nz = (double)N * H * W, nd = (double)N * oh * ow;
    const double mz = wsd[c * 4 + 0] / nz, md = wsd[c * 4 + 2] / nd;
    // unbiased std (torch.std_mean default), tilevae.py:553-554
    const float std_old = (float)sqrt(fmax(0.0, (wsd[c * 4 + 1] - nz * mz * mz) / (nz - 1.0)));
    const float std_new = (float)sqrt(fmax(0.0, (wsd[c * 4 + 3] - nd * md * md) / (nd - 1.0)));
    const float mean_old = (float)mz, mean_new = (float)md;
    float v = z[((size_t)n * C + c) * H * W + (size_t)nearest_exact_src(oy, sc_h, H) * W + nearest_exact_src(ox, sc_w, W)];
    v = (v - mean_new) / std_new * std_old + mean_old;  // tilevae.py:555
    v = fminf(fmaxf(v, zmin), zmax);                    // tilevae.py:559
    out[i] = v;
}

}  // namespace

namespace mdt {
// mean / var [B * groups] of a conv output [B, cout, H * W = HW] from the partials its epilogue left (d_cpart, see k_conv_stats_partial);
// d_gnws: mdtile_gn_stats_ws_size(B, groups) bytes
int conv_stats_finish_launch(const double* d_cpart, int B, int cout, size_t HW, int units, int NCB, int QB, int groups, float* d_mean, float* d_var,
                             void* d_gnws, hipStream_t s) {
    const int BG = B * groups, cpg = cout / groups;
    MDT_CHECK_ARG(groups > 0 && cout % groups == 0 && cpg % 4 == 0 && (4 * QB) % cpg == 0 && BG <= 65535,
                  "conv statistics: %d channels per group do not tile the kernel's %d-cout blocks in quads", cpg, 4 * QB);
    hipLaunchKernelGGL(k_conv_stats_partial, dim3(GN_NBLK, BG), dim3(256), 0, s, d_cpart, units, NCB, QB, cpg, groups, (double*)d_gnws);
    MDT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gn_final, dim3(cdiv(BG, 128)), dim3(128), 0, s, (const double*)d_gnws, (size_t)cpg * HW, BG, d_mean, d_var);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}
}  // namespace mdt

extern "C" size_t mdtile_gn_stats_ws_size(int B, int groups) { return (size_t)B * groups * GN_NBLK * 2 * sizeof(double); }

extern "C" int mdtile_gn_stats(const float* d_x, int B, int C, int HW, int groups, float* d_mean, float* d_var, void* d_ws,
                               mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_mean && d_var && d_ws, "mdtile_gn_stats: null argument");
    MDT_CHECK_ARG(B > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "mdtile_gn_stats: bad shape B=%d C=%d HW=%d groups=%d", B, C, HW, groups);
    const int BG = B * groups;
    MDT_CHECK_ARG(BG <= 65535, "mdtile_gn_stats: too many groups");
    const size_t L = (size_t)(C / groups) * HW;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_gn_partial, dim3(GN_NBLK, BG), dim3(256), 0, s, d_x, L, (double*)d_ws);
    MDT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gn_final, dim3(cdiv(BG, 64)), dim3(64), 0, s, (const double*)d_ws, L, BG, d_mean, d_var);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_gn_sums(const float* d_x, int B, int C, size_t plane_stride, size_t offset, size_t len, int groups, double* d_sums,
                              void* d_ws, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_sums && d_ws, "mdtile_gn_sums: null argument");
    MDT_CHECK_ARG(B > 0 && C > 0 && len > 0 && offset + len <= plane_stride && groups > 0 && C % groups == 0 && B * groups <= 65535,
                  "mdtile_gn_sums: bad shape B=%d C=%d groups=%d", B, C, groups);
    const int BG = B * groups;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_gn_partial_rows, dim3(GN_NBLK, BG), dim3(256), 0, s, d_x, C / groups, plane_stride, offset, len, (double*)d_ws);
    MDT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gn_sum_partials, dim3(cdiv(BG, 64)), dim3(64), 0, s, (const double*)d_ws, BG, d_sums);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_gn_from_sums(const double* d_sums, double count, int BG, float* d_mean, float* d_var, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_sums && d_mean && d_var && BG > 0 && count > 0.0, "mdtile_gn_from_sums: bad arguments");
    hipLaunchKernelGGL(k_gn_from_sums, dim3(cdiv(BG, 64)), dim3(64), 0, as_stream(stream), d_sums, count, BG, d_mean, d_var);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_gn_pool(const float* d_means, const float* d_vars, const float* d_frac, int T, int BG, float* d_mean,
                              float* d_var, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_means && d_vars && d_frac && d_mean && d_var && T > 0 && BG > 0, "mdtile_gn_pool: bad arguments");
    hipLaunchKernelGGL(k_gn_pool, dim3(cdiv(BG, 64)), dim3(64), 0, as_stream(stream), d_means, d_vars, d_frac, T, BG, d_mean, d_var);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_gn_apply(const float* d_x, float* d_y, int B, int C, int HW, int groups, const float* d_mean,
                               const float* d_var, const float* d_gamma, const float* d_beta, float eps, int silu,
                               mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_y && d_mean && d_var, "mdtile_gn_apply: null argument");
    MDT_CHECK_ARG(B > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0 && B * C <= 65535, "mdtile_gn_apply: bad shape");
    int gx = cdiv(cdiv(HW, 4), 256);
    if (gx > 64) gx = 64;  // grid-stride: ~planes*64 blocks keeps every CU busy without an oversized grid
    dim3 grid(gx, B * C), block(256);
    hipStream_t s = as_stream(stream);
    if (silu) hipLaunchKernelGGL(k_gn_apply<true>, grid, block, 0, s, d_x, d_y, C, HW, C / groups, groups, d_mean, d_var, d_gamma, d_beta, eps);
    else hipLaunchKernelGGL(k_gn_apply<false>, grid, block, 0, s, d_x, d_y, C, HW, C / groups, groups, d_mean, d_var, d_gamma, d_beta, eps);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

static int eltwise_grid(size_t n) {
    long long g = cdiv((long long)cdiv((long long)n, 4), 256);
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// Per-(sample, channel) affine of a fixed-statistics GroupNorm -- exactly the two constants k_gn_apply forms per plane:
// a = gamma / sqrt(var + eps), s = fma(-mean, a, beta).  Layout [B][2][C] (a row, then s row), consumed by mdtile_conv2d_gn.
__global__ void k_gn_coeffs(const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ gamma,
                            const float* __restrict__ beta, int B, int C, int cpg, int groups, float eps, float* __restrict__ coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    const int g = b * groups + c / cpg;
    const float rstd = 1.0f / sqrtf(var[g] + eps);
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    const float a = rstd * ga;
    coef[((size_t)b * 2 + 0) * C + c] = a;
    coef[((size_t)b * 2 + 1) * C + c] = fmaf(-mean[g], a, be);
}

extern "C" int mdtile_gn_coeffs(const float* d_mean, const float* d_var, const float* d_gamma, const float* d_beta, int B, int C, int groups,
                                float eps, float* d_coef, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_mean && d_var && d_coef, "mdtile_gn_coeffs: null argument");
    MDT_CHECK_ARG(B > 0 && C > 0 && groups > 0 && C % groups == 0, "mdtile_gn_coeffs: C=%d not divisible by groups=%d", C, groups);
    hipLaunchKernelGGL(k_gn_coeffs, dim3(cdiv((long long)B * C, 256)), dim3(256), 0, as_stream(stream), d_mean, d_var, d_gamma, d_beta, B, C,
                       C / groups, groups, eps, d_coef);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_silu(const float* d_x, float* d_y, size_t n, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_y, "mdtile_silu: null argument");
    if (n == 0) return MDTILE_OK;
    hipLaunchKernelGGL(k_eltwise<0>, dim3(eltwise_grid(n)), dim3(256), 0, as_stream(stream), d_x, (const float*)nullptr, d_y, n);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

// Decoder.tanh_out (ldm Decoder.forward; upstream queues it as the last task, scripts/tilevae.py:192-193)
extern "C" int mdtile_tanh(const float* d_x, float* d_y, size_t n, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_y, "mdtile_tanh: null argument");
    if (n == 0) return MDTILE_OK;
    hipLaunchKernelGGL(k_eltwise<2>, dim3(eltwise_grid(n)), dim3(256), 0, as_stream(stream), d_x, (const float*)nullptr, d_y, n);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_add(const float* d_a, const float* d_b, float* d_y, size_t n, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_a && d_b && d_y, "mdtile_add: null argument");
    if (n == 0) return MDTILE_OK;
    hipLaunchKernelGGL(k_eltwise<1>, dim3(eltwise_grid(n)), dim3(256), 0, as_stream(stream), d_a, d_b, d_y, n);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_crop_store(const float* d_tile, int N, int C, int th, int tw, const int* in_bbox4, const int* out_bbox4,
                                 int is_decoder, float* d_result, int RH, int RW, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_tile && d_result && in_bbox4 && out_bbox4 && N > 0 && C > 0 && N * C <= 65535, "mdtile_crop_store: bad arguments");
    int pad[4], m[4];
    for (int i = 0; i < 4; ++i) {
        pad[i] = is_decoder ? in_bbox4[i] * 8 : in_bbox4[i] / 8;  // tilevae.py:257
        m[i] = out_bbox4[i] - pad[i];                              // tilevae.py:258
    }
    // x[:, :, m2 : th + m3, m0 : tw + m1]  (tilevae.py:259)
    const int cw = tw + m[1] - m[0], ch = th + m[3] - m[2];
    MDT_CHECK_ARG(m[0] >= 0 && m[2] >= 0 && m[1] <= 0 && m[3] <= 0 && cw > 0 && ch > 0, "mdtile_crop_store: inconsistent bboxes (margins %d %d %d %d)", m[0], m[1], m[2], m[3]);
    MDT_CHECK_ARG(cw == out_bbox4[1] - out_bbox4[0] && ch == out_bbox4[3] - out_bbox4[2], "mdtile_crop_store: crop %dx%d != target window %dx%d", cw, ch,
                  out_bbox4[1] - out_bbox4[0], out_bbox4[3] - out_bbox4[2]);
    MDT_CHECK_ARG(out_bbox4[0] >= 0 && out_bbox4[2] >= 0 && out_bbox4[1] <= RW && out_bbox4[3] <= RH, "mdtile_crop_store: target window outside result");
    long long n = (long long)cw * ch;
    int gx = cdiv(n, 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_crop_store, dim3(gx, N * C), dim3(256), 0, as_stream(stream), d_tile, th, tw, m[0], m[2], cw, ch, d_result, RH, RW,
                       out_bbox4[0], out_bbox4[2]);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_vae_fast_size(int H, int W, int tile_size, int* oh, int* ow) {
    MDT_CHECK_ARG(H > 0 && W > 0 && tile_size > 0 && oh && ow, "mdtile_vae_fast_size: bad arguments");
    const double sf = (double)tile_size / (double)(H > W ? H : W);  // tilevae.py:545
    *oh = (int)floor((double)H * sf);                                // F.interpolate(scale_factor=...) output size
    *ow = (int)floor((double)W * sf);
    return MDTILE_OK;
}

extern "C" int mdtile_vae_fast_input(const float* d_z, int N, int C, int H, int W, int tile_size, float* d_out, void* d_ws,
                                     mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_z && d_out && d_ws && N > 0 && C > 0, "mdtile_vae_fast_input: bad arguments");
    int oh, ow;
    int rc = mdtile_vae_fast_size(H, W, tile_size, &oh, &ow);
    if (rc) return rc;
    MDT_CHECK_ARG(oh > 0 && ow > 0, "mdtile_vae_fast_input: empty estimator input");
    const double sf = (double)tile_size / (double)(H > W ? H : W);
    const float sc = (float)(1.0 / sf);  // torch: scale = 1/scale_factor when a scale_factor is given
    // workspace: [ per-channel totals: 4 doubles x C | partial doubles 4 x C x FAST_NBLK | totals min/max 2 floats x C | partial floats ]
    double* wsd = (double*)d_ws;
    double* pd = wsd + 4 * (size_t)C;
    float* wsf = (float*)(pd + 4 * (size_t)C * FAST_NBLK);
    float* pf = wsf + 2 * (size_t)C;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_fast_stats, dim3(FAST_NBLK, C), dim3(256), 0, s, d_z, N, C, H, W, oh, ow, sc, sc, pd, pf);
    MDT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fast_combine, dim3(cdiv(C, 64)), dim3(64), 0, s, (const double*)pd, (const float*)pf, C, wsd, wsf);
    MDT_LAUNCH_CHECK();
    size_t total = (size_t)N * C * oh * ow;
    hipLaunchKernelGGL(k_fast_apply, dim3(cdiv((long long)total, 256)), dim3(256), 0, s, d_z, N, C, H, W, d_out, oh, ow, sc, sc,
                       (const double*)wsd, (const float*)wsf);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" size_t mdtile_vae_fast_ws_size(int C) {
    return (size_t)C * (1 + FAST_NBLK) * (4 * sizeof(double) + 2 * sizeof(float));
}
