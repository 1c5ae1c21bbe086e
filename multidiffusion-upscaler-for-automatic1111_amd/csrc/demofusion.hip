// DemoFusion model evaluation (upstream tile_methods/demofusion.py:219-324, https://arxiv.org/abs/2311.16973).
// One hijacked evaluation =  local path: equally sized, randomly jittered windows -> count-averaged blend            (:244-257)
//                            global path: S x S dilated (strided) views of the Gaussian-filtered latent -> scatter     (:259-310)
//                            mix: x_local * (1 - c2) + x_global * c2                                                   (:311-322)
// Upstream runs ~3 T + 4 S^2 + 20 eager kernels per evaluation; here: one gather launch per tile batch family, one window-blend
// launch, one blur launch, one combine launch.  All gather-formulated (one thread owns an output element, no atomics, sums in
// upstream's list order).  I/O in the latent's dtype, fp32 accumulation.
#include "common.h"

using namespace mdt;

namespace {

// ---- local path: x_local = sum_{windows covering the pixel} out_w / max(count, 1) ------------------------------------------
// Windows are listed row-major over a rows x cols grid whose NOMINAL origins (nomx[c], nomy[r]) are jittered by at most +-J and then
// shifted by +J into the padded canvas: window (r, c) starts somewhere in [nom + 0, nom + 2J].  A pixel first rejects whole grid rows
// / columns by that bound, then tests the few candidates exactly.
template <typename T>
__global__ __launch_bounds__(256) void k_window_blend(const T* __restrict__ tiles, T* __restrict__ out, const int* __restrict__ wxy,
                                                      const int* __restrict__ nomx, const int* __restrict__ nomy, int rows, int cols, int J,
                                                      int win, int N, int C, int Hp, int Wp) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= Hp * Wp) return;
    const int y = px / Wp, x = px - y * Wp;
    const int plane = blockIdx.y;                       // n * C + c
    const size_t wplane = (size_t)win * win;
    float acc = 0.0f, cnt = 0.0f;
    for (int r = 0; r < rows; ++r) {
        const int ny = nomy[r];
        if (y < ny || y >= ny + 2 * J + win) continue;
        for (int c = 0; c < cols; ++c) {
            const int nx = nomx[c];
            if (x < nx || x >= nx + 2 * J + win) continue;
            const int w = r * cols + c;
            const int dx = x - wxy[2 * w], dy = y - wxy[2 * w + 1];
            if (dx < 0 || dy < 0 || dx >= win || dy >= win) continue;
            // tile-major batch rows: window w, sample n -> row w * N + n  (x_tile_out[i*N:(i+1)*N], :250)
            const int n = plane / C, ch = plane - n * C;
            acc += to_f32<T>(tiles[(((size_t)w * N + n) * C + ch) * wplane + (size_t)dy * win + dx]);
            cnt += 1.0f;
        }
    }
    out[(size_t)plane * Hp * Wp + px] = from_f32<T>(acc / (cnt == 0.0f ? 1.0f : cnt));     // weights == 0 -> 1 (:253)
}

// ---- global path, gather: cell (bx, by) of the S x S lattice -> x[:, :, by+J : end : S, bx+J : end : S]  (:268-283) -----------
constexpr int MAX_CELLS = 128;     // S <= 8, doubled in mixture mode
struct Cells {
    unsigned char bx[MAX_CELLS], by[MAX_CELLS];
    int n;
};

template <typename T>
__global__ __launch_bounds__(256) void k_dilated_gather(const T* __restrict__ a, const T* __restrict__ bsrc, int nfirst, T* __restrict__ out,
                                                        int N, int C, int Hp, int Wp, int S, int J, int h0, int w0, const Cells cells) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= h0 * w0) return;
    const int yy = idx / w0, xx = idx - yy * w0;
    const int ch = blockIdx.y;
    const int row = blockIdx.z, i = row / N, n = row - i * N;      // output row = cell i, sample n
    const T* src = i < nfirst ? a : bsrc;                           // mixture: the first cells view x_in, the rest the filtered latent
    out[((size_t)row * C + ch) * h0 * w0 + idx] =
        src[(((size_t)n * C + ch) * Hp + cells.by[i] + J + yy * S) * Wp + cells.bx[i] + J + xx * S];
}

// ---- scatter of the global outputs + mix with the local path (:284-322) ---------------------------------------------------------
//   x_global[strided view of cell i] += out_i   (cells in list order; the lattice cells partition the interior, mixture lists them twice)
//   x_global = (mixture ? x_global / 2 : x_global) / 1 ;   out = x_local * (1 - c2) + x_global * c2
template <typename T>
__global__ __launch_bounds__(256) void k_demofusion_combine(const T* __restrict__ x_local, const T* __restrict__ g, T* __restrict__ out, int N,
                                                            int C, int Hp, int Wp, int S, int J, int end, int h0, int w0, int mixture, float c2) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= Hp * Wp) return;
    const int y = px / Wp, x = px - y * Wp;
    const int plane = blockIdx.y, n = plane / C, ch = plane - n * C;
    float xg = 0.0f;
    if (x >= J && x < end && y >= J && y < end && y < Hp) {
        const int bx = (x - J) % S, by = (y - J) % S, xx = (x - J) / S, yy = (y - J) / S;
        if (yy < h0 && xx < w0) {
            const int cell = by * S + bx;
            const size_t off = (size_t)yy * w0 + xx;
            xg = to_f32<T>(g[(((size_t)cell * N + n) * C + ch) * h0 * w0 + off]);
            if (mixture) xg = (xg + to_f32<T>(g[(((size_t)(cell + S * S) * N + n) * C + ch) * h0 * w0 + off])) / 2.0f;
        }
    }
    const size_t o = (size_t)plane * Hp * Wp + px;
    out[o] = from_f32<T>(to_f32<T>(x_local[o]) * (1.0f - c2) + xg * c2);
}

// ---- Gaussian filter: depthwise K x K conv, zero padding (:173-178), optionally followed by the re-standardisation
//      (x_g - mean_g) / std_g * std_ + mean_ (:264) with the four scalars read from device memory ----------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_depthwise_blur(const T* __restrict__ x, const float* __restrict__ kern, T* __restrict__ out, int H, int W,
                                                        int K) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= H * W) return;
    const int y = px / W, xx = px - y * W, r = K / 2;
    const T* src = x + (size_t)blockIdx.y * H * W;
    float acc = 0.0f;
    for (int ky = 0; ky < K; ++ky) {
        const int sy = y + ky - r;
        if (sy < 0 || sy >= H) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int sx = xx + kx - r;
            if (sx < 0 || sx >= W) continue;
            acc = fmaf(to_f32<T>(src[(size_t)sy * W + sx]), kern[ky * K + kx], acc);
        }
    }
    out[(size_t)blockIdx.y * H * W + px] = from_f32<T>(acc);
}

template <typename T>
__global__ __launch_bounds__(256) void k_restandardize(const T* __restrict__ x, const float* __restrict__ st, T* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // st = { mean of x, std of x, target mean, target std }
    out[i] = from_f32<T>((to_f32<T>(x[i]) - st[0]) / st[1] * st[3] + st[2]);
}

template <typename F>
int by_dtype(int dtype, F&& f) {
    switch (dtype) {
        case MDTILE_DT_F32: f((float*)nullptr); return MDTILE_OK;
        case MDTILE_DT_F16: f((__half*)nullptr); return MDTILE_OK;
        case MDTILE_DT_BF16: f((__hip_bfloat16*)nullptr); return MDTILE_OK;
    }
    mdt::set_error("bad dtype %d", dtype);
    return MDTILE_E_ARG;
}

}  // namespace

extern "C" int mdtile_window_blend(int dtype, const void* d_tiles, void* d_out, const int* d_window_xy, const int* d_nomx, const int* d_nomy,
                                   int rows, int cols, int jitter, int window, int N, int C, int Hp, int Wp, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_tiles && d_out && d_window_xy && d_nomx && d_nomy, "mdtile_window_blend: null argument");
    MDT_CHECK_ARG(rows > 0 && cols > 0 && jitter >= 0 && window > 0 && N > 0 && C > 0 && N * C <= 65535 && Hp > 0 && Wp > 0,
                  "mdtile_window_blend: bad shape");
    hipStream_t s = as_stream(stream);
    int rc = by_dtype(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        hipLaunchKernelGGL(k_window_blend<T>, dim3(cdiv((long long)Hp * Wp, 256), N * C), dim3(256), 0, s, (const T*)d_tiles, (T*)d_out, d_window_xy, d_nomx,
                           d_nomy, rows, cols, jitter, window, N, C, Hp, Wp);
    });
    if (rc != MDTILE_OK) return rc;
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_dilated_gather(int dtype, const void* d_x, const void* d_x_filtered, int num_from_x, void* d_out, const int* cells_xy,
                                     int num_cells, int N, int C, int Hp, int Wp, int S, int jitter, int h0, int w0, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_out && cells_xy && num_cells > 0 && num_cells <= MAX_CELLS, "mdtile_dilated_gather: bad arguments (%d cells)", num_cells);
    MDT_CHECK_ARG(num_from_x >= 0 && num_from_x <= num_cells && (num_from_x == num_cells || d_x_filtered), "mdtile_dilated_gather: filtered source missing");
    MDT_CHECK_ARG(N > 0 && C > 0 && C <= 65535 && S > 0 && jitter >= 0 && h0 > 0 && w0 > 0 && (long long)num_cells * N <= 65535, "mdtile_dilated_gather: bad shape");
    Cells cl;
    cl.n = num_cells;
    for (int i = 0; i < num_cells; ++i) {
        const int bx = cells_xy[2 * i], by = cells_xy[2 * i + 1];
        MDT_CHECK_ARG(bx >= 0 && by >= 0 && bx < S && by < S && by + jitter + (h0 - 1) * S < Hp && bx + jitter + (w0 - 1) * S < Wp,
                      "mdtile_dilated_gather: cell %d (%d,%d) leaves the canvas", i, bx, by);
        cl.bx[i] = (unsigned char)bx;
        cl.by[i] = (unsigned char)by;
    }
    hipStream_t s = as_stream(stream);
    int rc = by_dtype(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        hipLaunchKernelGGL(k_dilated_gather<T>, dim3(cdiv((long long)h0 * w0, 256), C, num_cells * N), dim3(256), 0, s, (const T*)d_x,
                           (const T*)(d_x_filtered ? d_x_filtered : d_x), num_from_x, (T*)d_out, N, C, Hp, Wp, S, jitter, h0, w0, cl);
    });
    if (rc != MDTILE_OK) return rc;
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_demofusion_combine(int dtype, const void* d_x_local, const void* d_global_out, void* d_out, int N, int C, int Hp, int Wp, int S,
                                         int jitter, int h0, int w0, int mixture, float c2, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x_local && d_global_out && d_out, "mdtile_demofusion_combine: null argument");
    MDT_CHECK_ARG(N > 0 && C > 0 && N * C <= 65535 && Hp > 0 && Wp > 0 && S > 0 && jitter >= 0 && h0 > 0 && w0 > 0, "mdtile_demofusion_combine: bad shape");
    hipStream_t s = as_stream(stream);
    const int end = Wp - jitter;      // upstream takes the end of BOTH axes from the width (:262)
    int rc = by_dtype(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        hipLaunchKernelGGL(k_demofusion_combine<T>, dim3(cdiv((long long)Hp * Wp, 256), N * C), dim3(256), 0, s, (const T*)d_x_local, (const T*)d_global_out,
                           (T*)d_out, N, C, Hp, Wp, S, jitter, end, h0, w0, mixture, c2);
    });
    if (rc != MDTILE_OK) return rc;
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_depthwise_blur(int dtype, const void* d_x, const float* d_kernel, void* d_out, int planes, int H, int W, int K,
                                     mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_kernel && d_out && planes > 0 && planes <= 65535 && H > 0 && W > 0 && K > 0 && (K & 1), "mdtile_depthwise_blur: bad arguments");
    hipStream_t s = as_stream(stream);
    int rc = by_dtype(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        hipLaunchKernelGGL(k_depthwise_blur<T>, dim3(cdiv((long long)H * W, 256), planes), dim3(256), 0, s, (const T*)d_x, d_kernel, (T*)d_out, H, W, K);
    });
    if (rc != MDTILE_OK) return rc;
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

extern "C" int mdtile_restandardize(int dtype, const void* d_x, const float* d_stats4, void* d_out, size_t n, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_stats4 && d_out && n > 0, "mdtile_restandardize: bad arguments");
    hipStream_t s = as_stream(stream);
    int rc = by_dtype(dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        hipLaunchKernelGGL(k_restandardize<T>, dim3(cdiv((long long)n, 256)), dim3(256), 0, s, (const T*)d_x, d_stats4, (T*)d_out, n);
    });
    if (rc != MDTILE_OK) return rc;
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}
