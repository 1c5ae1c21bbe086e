// Conv tiles of the Tiled-VAE task queue (K15, K16): implicit-GEMM 3x3 / 1x1 convolution on the fp32 matrix cores.
// Upstream runs these as nn.Conv2d modules held by the queue (scripts/tilevae.py:115-195: conv_in, conv1, conv2,
// nin_shortcut, upsample.conv, conv_out, and q/k/v/proj_out of tile_utils/attn.py:50-70), plus a separate in-place
// residual add (tilevae.py:614-616) and F.interpolate(nearest, 2x) before the upsample conv.
//
// Design (gfx950):
//   * exact fp32: v_mfma_f32_32x32x2_f32 (bit-identical to an fmaf chain; 157 TFLOP/s peak, no TF32 on CDNA4)
//   * NCHW in, NCHW (or token-major) out.  GEMM view:  D[cout][px] = sum_{tap,cin} Wp[tap][cin][cout] * X[cin][px+tap]
//     M = 32 output channels per MFMA, N = 32 consecutive pixels of one image row, K = 2 input channels of one tap.
//     Both operands are K-major in LDS, so every ds_read_b32 is a conflict-free 32-lane run.
//   * block = 4 waves, output tile 8 rows x 32 px x BN channels; the (8+2)x(32+2) input halo of a KC-channel slab is
//     staged once in LDS and re-read by all 9 taps (9x on-chip reuse) - global traffic is 1 read of X per BN-block.
//   * register-prefetch double buffering: the next slab's global loads are issued before the current slab's MFMAs and
//     written to the other LDS buffer after them: one barrier per slab.
//   * nearest-2x upsample and the residual add are fused (source index >> 1 while staging; epilogue add).
//   * XCD-aware block order: the BN-blocks of one pixel tile sit on the same XCD (same L2) back to back.
#include "common.h"
#include <algorithm>

using namespace mdt;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct ConvParams {
    const float* x;         // [B, Cin, Hin, Win]
    const float* w;         // packed [KS*KS][Cin][CoutP]
    const float* bias;      // [Cout] or null
    const float* res;       // residual, same layout as y, or null
    float* y;
    int B, Cin, Cout, CoutP, H, W;  // H, W: OUTPUT spatial size
    int Hin, Win, up;               // input spatial size; up = 1 for fused nearest-2x
    int ptiles, PX, NCB;
};

// S = 2: ldm's Downsample (F.pad(x, (0,1,0,1)) then a stride-2 3x3 conv without padding; the encoder's 'downsample' task,
// scripts/tilevae.py:155-171): out[y][x] = sum w[dy][dx] in[2y+dy][2x+dx], zero beyond the last input row / column.
template <int KS, int KC, int WP, int WC, int RP, int RC, bool TOKMAJ, int S = 1, int MINB = 1>
__global__ __launch_bounds__(256, MINB) void k_conv(const ConvParams P) {
    static_assert(WP * WC == 4 && WP * RP == 8, "4 waves cover 8 rows");
    constexpr int PAD = S == 1 ? KS / 2 : 0, ROWS = 7 * S + KS, TWP = 31 * S + KS, TAPS = KS * KS, BN = WC * RC * 32;
    constexpr int E_IN = KC * ROWS * TWP, E_IN_P = (E_IN + 3) & ~3, E_WT = TAPS * KC * BN;
    constexpr int NI = (E_IN + 255) / 256, NW = (E_WT / 4 + 255) / 256, BUF = E_IN_P + E_WT;
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

    // ---- block -> (pixel tile, channel block): XCD = id % 8 keeps all NCB channel blocks of a pixel tile on one L2
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int ptile = (slot / P.NCB) * 8 + xcd, cb = slot % P.NCB;
    if (ptile >= P.ptiles) return;
    const int b = blockIdx.y;
    const int py = ptile / P.PX, px = ptile - py * P.PX;
    const int y0 = py * 8, x0 = px * 32, n0 = cb * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int wp = wave / WC, wc = wave % WC;
    const size_t HWin = (size_t)P.Hin * P.Win;
    const float* xb = P.x + (size_t)b * P.Cin * HWin;

    // ---- stage-independent staging maps
    int goff[NI];  // offset inside a channel plane, or -1 (zero padding / outside); the channel kc of slot i is (tid + 256 i) / (ROWS TWP)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = tid + 256 * i;
        goff[i] = -1;
        if (idx < E_IN) {
            const int kc = idx / (ROWS * TWP), rem = idx - kc * (ROWS * TWP);
            const int r = rem / TWP, c = rem - r * TWP;
            const int gy = y0 * S + r - PAD, gx = x0 * S + c - PAD;
            const int hlim = S == 1 ? P.H : P.Hin, wlim = S == 1 ? P.W : P.Win;   // S = 1: bounds of the (possibly upsampled) input view
            if (gy >= 0 && gy < hlim && gx >= 0 && gx < wlim) {
                const int sy = P.up ? gy >> 1 : gy, sx = P.up ? gx >> 1 : gx;
                goff[i] = sy * P.Win + sx;
            }
        }
    }
    float rin[NI];
    float4 rwt[NW];

    auto load_stage = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int g = goff[i];
            const int kc = (tid + 256 * i) / (ROWS * TWP);
            rin[i] = (g >= 0 && c0 + kc < P.Cin) ? xb[(size_t)(c0 + kc) * HWin + g] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int q = tid + 256 * i;  // float4 index in [tap][kc][BN/4]
            const int n4 = q % (BN / 4), kc = (q / (BN / 4)) % KC, tap = q / (BN / 4) / KC;
            const int cin = c0 + kc, n = n0 + 4 * n4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < E_WT / 4 && cin < P.Cin && n < P.CoutP)
                v = *reinterpret_cast<const float4*>(P.w + ((size_t)tap * P.Cin + cin) * P.CoutP + n);
            rwt[i] = v;
        }
    };
    auto store_stage = [&](int buf) {
        float* in_l = smem + buf * BUF;
        float* wt_l = in_l + E_IN_P;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = tid + 256 * i;
            if (idx < E_IN) in_l[idx] = rin[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int q = tid + 256 * i;
            if (q < E_WT / 4) reinterpret_cast<float4*>(wt_l)[q] = rwt[i];
        }
    };

    f32x16 acc[RP][RC];
#pragma unroll
    for (int r = 0; r < RP; ++r)
#pragma unroll
        for (int j = 0; j < RC; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[r][j][q] = 0.0f;

    const int nst = (P.Cin + KC - 1) / KC;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        if (st + 1 < nst) load_stage((st + 1) * KC);
        const float* in_l = smem + (st & 1) * BUF;
        const float* wt_l = in_l + E_IN_P;
        // lane view: hi selects the k of the pair, l31 the pixel / the output channel
        const float* inb = in_l + hi * (ROWS * TWP) + (wp * RP) * S * TWP + l31 * S;
        const float* wtb = wt_l + hi * BN + wc * (RC * 32) + l31;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = tap / KS, dx = tap % KS;
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) {
                float wv[RC], xv[RP];
#pragma unroll
                for (int j = 0; j < RC; ++j) wv[j] = wtb[(tap * KC + 2 * s) * BN + j * 32];
#pragma unroll
                for (int r = 0; r < RP; ++r) xv[r] = inb[(2 * s * ROWS + r * S + dy) * TWP + dx];
#pragma unroll
                for (int r = 0; r < RP; ++r)
#pragma unroll
                    for (int j = 0; j < RC; ++j) {
                        if (TOKMAJ) acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[r], wv[j], acc[r][j], 0, 0, 0);
                        else        acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j], xv[r], acc[r][j], 0, 0, 0);
                    }
            }
        }
        if (st + 1 < nst) store_stage((st + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: + bias (+ residual), store.  C/D layout of 32x32 MFMA: col = lane & 31, row = (q&3) + 8*(q>>2) + 4*hi
    const size_t HW = (size_t)P.H * P.W;
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        const int y = y0 + wp * RP + r;
        if (y >= P.H) continue;
#pragma unroll
        for (int j = 0; j < RC; ++j) {
            const int cbase = n0 + (wc * RC + j) * 32;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * hi;
                if (!TOKMAJ) {  // rows = output channel, cols = pixel  ->  128-B runs along x
                    const int co = cbase + row, x = x0 + l31;
                    if (co < P.Cout && x < P.W) {
                        const size_t o = ((size_t)b * P.Cout + co) * HW + (size_t)y * P.W + x;
                        float v = acc[r][j][q] + (P.bias ? P.bias[co] : 0.0f);
                        if (P.res) v += P.res[o];
                        P.y[o] = v;
                    }
                } else {        // rows = pixel, cols = output channel  ->  128-B runs along c of [B, HW, Cout]
                    const int co = cbase + l31, x = x0 + row;
                    if (co < P.Cout && x < P.W) {
                        const size_t o = ((size_t)b * HW + (size_t)y * P.W + x) * P.Cout + co;
                        float v = acc[r][j][q] + (P.bias ? P.bias[co] : 0.0f);
                        if (P.res) v += P.res[o];
                        P.y[o] = v;
                    }
                }
            }
        }
    }
}

__global__ void k_conv_pack(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int KS, int CoutP) {
    const size_t n = (size_t)KS * KS * Cin * CoutP;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int co = (int)(i % CoutP);
    const int ci = (int)((i / CoutP) % Cin);
    const int tap = (int)(i / ((size_t)CoutP * Cin));
    wp[i] = co < Cout ? w[((size_t)co * Cin + ci) * KS * KS + tap] : 0.0f;  // OIHW, tap = ky*KS + kx
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- conv_in (round 5): 3x3 convs with 3 or 4 input channels (the encoder's first conv 3 -> 128 on the IMAGE tile, the decoder's 4 -> 512 on the
// latent tile; scripts/tilevae.py:115-137 'conv_in').  K = 27 / 36 leaves the matrix cores nothing to do -- the MFMA kernel above ran it at
// 14 TFLOP/s, 4.3 ms per 3072^2 encoder tile -- and the op is its OUTPUT stream: 128 planes x 4 B per pixel.  Plain fp32 FMAs (exact fp32 in
// every precision mode): a lane owns two adjacent pixels of a row as packed pairs (v_pk_fma_f32), its 4 x 3 x CIN input values live in
// registers, the weights of 128 couts sit in LDS [cout][tap][cin] and are read as wave-wide broadcasts (16 B per read, no conflicts);
// one 8-byte store per cout and lane = 512-byte runs per wave.  Block = 4 waves = 4 rows x 128 px x 128 couts.
//
// Round 6: every weight sits in LDS TWICE, as the pair (w, w) the packed FMA consumes, so that no operand of the loop needs an op_sel
// modifier.  The round-5 form kept one copy, read quads (w0 w1 w2 w3) and let hipcc broadcast each to both halves of the pair; for w1 that is
// `v_pk_fma_f32 ... op_sel:[0,1,0]` (the LOW half of the product reads the HIGH register of the source pair).  Alone on the GPU that kernel was
// bit-stable over every test of round 5; sharing its CU with another kernel's MFMA waves (another process on the same GPU, or another stream
// of this one) it returned, in 43-100 % of its calls, values in which exactly that one product was missing -- low half only, lanes 48-63
// only, a few couts per wave (probes/contention_fewcin.py, profiles/r6j: the dump of one such call is explained term by term).  The LDS
// reads, the registers, the stores and the FMAs of every other encoding passed the same test (probes/contention_regkeep.py; forms 2-5 of the
// investigation).  FORM 1 (PROBES twin, MDTILE_FEWCIN_FORM=1) is the round-5 kernel, kept so that the finding stays reproducible;
// tools/asm_guard.py checks that the shipping kernel's packed FMAs carry no op_sel.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int CIN, int FORM = 0>
__global__ __launch_bounds__(256) void k_conv3x3_fewcin(const float* __restrict__ x, const float* __restrict__ wpk, const float* __restrict__ bias,
                                                        const float* __restrict__ res, float* __restrict__ y, int Cout, int CoutP, int NCB, int H, int W) {
    constexpr int K = 9 * CIN, KP = (K + 3) & ~3;
    constexpr int DUP = FORM == 0 ? 2 : 1;                 // copies of each weight in LDS
    __shared__ float4 wl4[128 * KP * DUP / 4];
    float* const wl = reinterpret_cast<float*>(wl4);
    const int cb = blockIdx.z % NCB, b = blockIdx.z / NCB;
    for (int i = threadIdx.x; i < 128 * KP; i += 256) {       // packed fp32 image [tap][cin][CoutP] -> LDS [cout][tap * CIN + cin] (x DUP)
        const int co = i & 127, k = i >> 7;
        const float wv = (k < K && cb * 128 + co < Cout) ? wpk[(size_t)k * CoutP + cb * 128 + co] : 0.0f;
        if constexpr (DUP == 2) reinterpret_cast<f32x2*>(wl)[co * KP + k] = f32x2{wv, wv};
        else wl[co * KP + k] = wv;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int yy = blockIdx.y * 4 + wave, x0 = blockIdx.x * 128 + lane * 2;
    if (yy >= H || x0 >= W) return;
    const size_t HW = (size_t)H * W;
    const float* xb = x + (size_t)b * CIN * HW;
    f32x2 in[KP];      // in[tap * CIN + ci] = (pixel x0, pixel x0 + 1) under tap (dy, dx): input columns x0 + dx - 1, x0 + dx
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int r = yy + dy - 1;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int xc = x0 + c - 1;
                v[c] = (r >= 0 && r < H && xc >= 0 && xc < W) ? xb[(size_t)ci * HW + (size_t)r * W + xc] : 0.0f;      // 'same' zero padding
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) in[(dy * 3 + dx) * CIN + ci] = f32x2{v[dx], v[dx + 1]};
        }
#pragma unroll
    for (int k = K; k < KP; ++k) in[k] = f32x2{0.0f, 0.0f};
    const bool two = x0 + 1 < W;
    const size_t o0 = (size_t)yy * W + x0;
    const bool pair_ok = two && ((o0 & 1) == 0) && ((HW & 1) == 0);      // 8-byte aligned in every plane
    const int nco = Cout - cb * 128 < 128 ? Cout - cb * 128 : 128;
    float* yb = y + ((size_t)b * Cout + cb * 128) * HW + o0;
    const float* rb = res ? res + ((size_t)b * Cout + cb * 128) * HW + o0 : nullptr;
#pragma unroll 4
    for (int co = 0; co < nco; ++co) {      // (four couts in flight: independent FMA chains)
        const float bv = bias ? bias[cb * 128 + co] : 0.0f;
        f32x2 acc = f32x2{0.0f, 0.0f};
        if constexpr (DUP == 2) {
            const float4* wr = wl4 + co * (KP / 2);
#pragma unroll
            for (int k2 = 0; k2 < KP / 2; ++k2) {
                const float4 w = wr[k2];      // (w[2 k2], w[2 k2], w[2 k2 + 1], w[2 k2 + 1]): two ready-made pairs
                acc = __builtin_elementwise_fma(in[2 * k2 + 0], f32x2{w.x, w.y}, acc);
                acc = __builtin_elementwise_fma(in[2 * k2 + 1], f32x2{w.z, w.w}, acc);
            }
        } else {
            const float4* wr = wl4 + co * (KP / 4);
#pragma unroll
            for (int k4 = 0; k4 < KP / 4; ++k4) {
                const float4 w = wr[k4];
                acc = __builtin_elementwise_fma(in[4 * k4 + 0], f32x2{w.x, w.x}, acc);
                acc = __builtin_elementwise_fma(in[4 * k4 + 1], f32x2{w.y, w.y}, acc);
                acc = __builtin_elementwise_fma(in[4 * k4 + 2], f32x2{w.z, w.z}, acc);
                acc = __builtin_elementwise_fma(in[4 * k4 + 3], f32x2{w.w, w.w}, acc);
            }
        }
        acc += f32x2{bv, bv};
        float* yp = yb + (size_t)co * HW;
        if (rb) {
            acc.x += rb[(size_t)co * HW];
            if (two) acc.y += rb[(size_t)co * HW + 1];
        }
        if (pair_ok) {
            *reinterpret_cast<f32x2*>(yp) = acc;
        } else {
            yp[0] = acc.x;
            if (two) yp[1] = acc.y;
        }
    }
}

static bool conv_fewcin_eligible(int cin, int ksize, int up, int out_layout) { return ksize == 3 && (cin == 3 || cin == 4) && !up && out_layout == 0; }

template <int KS, int KC, int WP, int WC, int RP, int RC, int MINB = 1>
int launch_conv(ConvParams& P, int out_layout, hipStream_t s) {
    constexpr int BN = WC * RC * 32;
    P.PX = (P.W + 31) / 32;
    P.ptiles = P.PX * ((P.H + 7) / 8);
    P.NCB = (P.CoutP + BN - 1) / BN;
    dim3 grid(((P.ptiles + 7) / 8) * 8 * P.NCB, P.B), block(256);
    if (out_layout == 1) hipLaunchKernelGGL((k_conv<KS, KC, WP, WC, RP, RC, true, 1, MINB>), grid, block, 0, s, P);
    else hipLaunchKernelGGL((k_conv<KS, KC, WP, WC, RP, RC, false, 1, MINB>), grid, block, 0, s, P);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

template <int KC, int WP, int WC, int RP, int RC>
int launch_conv_down2(ConvParams& P, hipStream_t s) {
    constexpr int BN = WC * RC * 32;
    P.PX = (P.W + 31) / 32;
    P.ptiles = P.PX * ((P.H + 7) / 8);
    P.NCB = (P.CoutP + BN - 1) / BN;
    dim3 grid(((P.ptiles + 7) / 8) * 8 * P.NCB, P.B), block(256);
    hipLaunchKernelGGL((k_conv<3, KC, WP, WC, RP, RC, false, 2>), grid, block, 0, s, P);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

}  // namespace

// vae_conv_bf16x3.hip
namespace mdt {
bool conv_bf16x3_eligible(int cout, int cin, int ksize);
size_t conv_bf16x3_packed_floats(int cout, int cin);
int conv_bf16x3_pack(const float* d_w_oihw, void* d_out, int cout, int cin, hipStream_t s);
bool conv_bf16x3_gn_supported(int cout, int cin, int ksize, int up);
int conv_bf16x3_down2_launch(const float* d_x, const void* d_w_rec, const float* d_bias, float* d_y, int B, int cin, int cout, int Hin, int Win,
                             hipStream_t s);
int conv_bf16x3_launch(const float* d_x, const void* d_w_rec, const float* d_bias, const float* d_res, float* d_y, int B, int cin,
                       int cout, int H, int W, int up, const float* d_coef, hipStream_t s, double* d_part = nullptr);
bool conv_bf16x3_stats_supported(int cout, int cin, int ksize, int up);
size_t conv_bf16x3_stats_part_doubles(int B, int cout, int H, int W);
// vae_norm.hip
int conv_stats_finish_launch(const double* d_cpart, int B, int cout, size_t HW, int units, int NCB, int QB, int groups, float* d_mean, float* d_var,
                             void* d_gnws, hipStream_t s);
bool conv_rec_narrow_eligible(int cout, int cin, int ksize);
size_t conv_rec_narrow_packed_floats(int cin);
int conv_rec_narrow_pack(const float* d_w_oihw, void* d_out, int cout, int cin, hipStream_t s);
// vae_conv_rec.hip
bool conv_rec_supported(int cout, int cin, int ksize);
size_t rec_image_bytes(int B, int C, int H, int W);
size_t rec_plane_records(int H, int W);
int rec_from_f32_launch(const float* d_x, const float* d_coef, void* d_rec, int B, int C, int H, int W, hipStream_t s);
int rec_to_f32_launch(const void* d_rec, float* d_x, int B, int C, int H, int W, hipStream_t s);
int conv_rec_launch(const void* d_xrec, const void* d_w_rec, const float* d_bias, const float* d_res, float* d_y32, void* d_yrec,
                    const float* d_ycoef, int B, int cin, int cout, int H, int W, int up, hipStream_t s, const int* win = nullptr, int family = 0,
                    double* d_part = nullptr);
bool conv_rec_stats_in_epilogue(int B, int cin, int cout, int H, int W, int up);
int conv_rec_stats_units(int H, int W, int up);
// vae_conv1x1_bf16x3.hip
bool conv1x1_bf16x3_eligible(int cout, int cin);
size_t conv1x1_bf16x3_packed_floats(int cout, int cin);
int conv1x1_bf16x3_pack(const float* d_w_oihw, void* d_out, int cout, int cin, hipStream_t s);
int conv1x1_bf16x3_launch(const float* d_x, const void* d_w_rec, const float* d_bias, const float* d_res, float* d_y, int B, int cin,
                          int cout, size_t HW, hipStream_t s);
}  // namespace mdt

// packed buffer = [ fp32 image (tap, cin, coutP) | split-bf16 record image (only for shapes the bf16x3 kernels take) ]
static size_t f32_packed_floats(int cout, int cin, int ksize) { return ((size_t)ksize * ksize * cin * round_up(cout, 32) + 3) & ~(size_t)3; }

extern "C" size_t mdtile_conv_packed_size(int cout, int cin, int ksize) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return 0;
    if (ksize == 1) return f32_packed_floats(cout, cin, ksize) + (conv1x1_bf16x3_eligible(cout, cin) ? conv1x1_bf16x3_packed_floats(cout, cin) : 0);
    return f32_packed_floats(cout, cin, ksize) + (conv_bf16x3_eligible(cout, cin, ksize) ? conv_bf16x3_packed_floats(cout, cin)
                                                  : conv_rec_narrow_eligible(cout, cin, ksize) ? conv_rec_narrow_packed_floats(cin) : 0);
}

extern "C" int mdtile_conv_pack(const float* d_w_oihw, float* d_w_packed, int cout, int cin, int ksize, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_w_oihw && d_w_packed && cout > 0 && cin > 0 && (ksize == 1 || ksize == 3), "mdtile_conv_pack: bad arguments");
    const int CoutP = round_up(cout, 32);
    const size_t n = (size_t)ksize * ksize * cin * CoutP;
    hipLaunchKernelGGL(k_conv_pack, dim3(cdiv((long long)n, 256)), dim3(256), 0, as_stream(stream), d_w_oihw, d_w_packed, cout, cin, ksize, CoutP);
    MDT_LAUNCH_CHECK();
    if (ksize == 1 && conv1x1_bf16x3_eligible(cout, cin))
        return conv1x1_bf16x3_pack(d_w_oihw, d_w_packed + f32_packed_floats(cout, cin, ksize), cout, cin, as_stream(stream));
    if (conv_bf16x3_eligible(cout, cin, ksize))
        return conv_bf16x3_pack(d_w_oihw, d_w_packed + f32_packed_floats(cout, cin, ksize), cout, cin, as_stream(stream));
    if (conv_rec_narrow_eligible(cout, cin, ksize))
        return conv_rec_narrow_pack(d_w_oihw, d_w_packed + f32_packed_floats(cout, cin, ksize), cout, cin, as_stream(stream));
    return MDTILE_OK;
}

extern "C" int mdtile_conv2d(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual, float* d_y,
                             int B, int cin, int cout, int H, int W, int ksize, int flags, int out_layout, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_w_packed && d_y, "mdtile_conv2d: null argument");
    MDT_CHECK_ARG(B > 0 && B <= 65535 && cin > 0 && cout > 0 && H > 0 && W > 0, "mdtile_conv2d: bad shape B=%d cin=%d cout=%d H=%d W=%d", B, cin, cout, H, W);
    MDT_CHECK_ARG(ksize == 1 || ksize == 3, "mdtile_conv2d: ksize %d unsupported (1 or 3)", ksize);
    MDT_CHECK_ARG(out_layout == 0 || out_layout == 1, "mdtile_conv2d: bad out_layout %d", out_layout);
    const int up = (flags & MDTILE_CONV_UPSAMPLE2X) ? 1 : 0;
    MDT_CHECK_ARG(!up || (H % 2 == 0 && W % 2 == 0), "mdtile_conv2d: upsample2x needs even output size, got %dx%d", H, W);
    MDT_CHECK_ARG((size_t)(up ? H / 2 : H) * (up ? W / 2 : W) < (1u << 31), "mdtile_conv2d: input plane of 2^31 or more pixels");
    ConvParams P;
    P.x = d_x; P.w = d_w_packed; P.bias = d_bias; P.res = d_residual; P.y = d_y;
    P.B = B; P.Cin = cin; P.Cout = cout; P.CoutP = round_up(cout, 32); P.H = H; P.W = W;
    P.Hin = up ? H / 2 : H; P.Win = up ? W / 2 : W; P.up = up;
    hipStream_t s = as_stream(stream);
    // split-bf16 matrix-core path (16/3 x the fp32-MFMA rate, ~1e-5 relative): default for the 3x3 convs it covers
    const bool force_f32 = conv_strict_f32();
    if (!force_f32 && !(flags & MDTILE_CONV_EXACT_F32) && out_layout == 0 && conv_bf16x3_eligible(cout, cin, ksize))
        return conv_bf16x3_launch(d_x, d_w_packed + f32_packed_floats(cout, cin, ksize), d_bias, d_residual, d_y, B, cin, cout, H, W, up, nullptr, s);
    // 1x1 convs (nin_shortcut, q / k / proj_out): split-bf16 kernel over the flat pixel run
    if (ksize == 1 && !up && !force_f32 && !(flags & MDTILE_CONV_EXACT_F32) && out_layout == 0 && conv1x1_bf16x3_eligible(cout, cin))
        return conv1x1_bf16x3_launch(d_x, d_w_packed + f32_packed_floats(cout, cin, ksize), d_bias, d_residual, d_y, B, cin, cout,
                                     (size_t)H * W, s);
    if (conv_fewcin_eligible(cin, ksize, up, out_layout)) {      // conv_in: exact fp32 FMAs, bound by its output stream (every precision mode)
        const int ncb = (cout + 127) / 128;
        MDT_CHECK_ARG((size_t)B * ncb <= 65535 && (H + 3) / 4 <= 65535, "mdtile_conv2d: conv_in grid too large (B=%d cout=%d H=%d)", B, cout, H);
        dim3 grid((W + 127) / 128, (H + 3) / 4, B * ncb), block(256);
        if constexpr (kProbes) {      // PROBES twin, MDTILE_FEWCIN_FORM=1: the round-5 kernel (one copy of the weights, op_sel broadcasts; see above)
            if (const char* e = probe_env("MDTILE_FEWCIN_FORM"); e && atoi(e) == 1) {
                if (cin == 3) hipLaunchKernelGGL((k_conv3x3_fewcin<3, 1>), grid, block, 0, s, d_x, d_w_packed, d_bias, d_residual, d_y, cout, P.CoutP, ncb, H, W);
                else hipLaunchKernelGGL((k_conv3x3_fewcin<4, 1>), grid, block, 0, s, d_x, d_w_packed, d_bias, d_residual, d_y, cout, P.CoutP, ncb, H, W);
                MDT_LAUNCH_CHECK();
                return MDTILE_OK;
            }
        }
        if (cin == 3) hipLaunchKernelGGL((k_conv3x3_fewcin<3>), grid, block, 0, s, d_x, d_w_packed, d_bias, d_residual, d_y, cout, P.CoutP, ncb, H, W);
        else hipLaunchKernelGGL((k_conv3x3_fewcin<4>), grid, block, 0, s, d_x, d_w_packed, d_bias, d_residual, d_y, cout, P.CoutP, ncb, H, W);
        MDT_LAUNCH_CHECK();
        return MDTILE_OK;
    }
    const bool wide = P.CoutP > 64;
    if (ksize == 3) {
        // Block shape of the wide exact-fp32 conv (round 6, probes/convf32_probe.py, profiles/r6l): 8-channel slabs cost 95 KB of LDS and 288
        // registers = ONE 4-wave block per CU; 2-channel slabs (24 KB, 188 registers under __launch_bounds__(256, 2)) run two blocks per
        // CU and hide each other's barriers: 116 -> 126 TF at 256 channels / 556^2, 108 -> 123 TF at 128 / 1112^2, 120 -> 129 TF at
        // 512 -> 256 and at four stacked 512 -> 512 tiles of 278^2 (the decode's stacking depth).  ONE such tile alone loses (118 -> 105 TF:
        // its 1260 blocks fill 512 slots 2.46 times); the rule does not look at the batch, so a tile decoded in a stack and alone takes
        // the same kernel and the same summation order.
        int form = 3;
        if constexpr (kProbes) {
            if (const char* e = probe_env("MDTILE_CONVF32_FORM")) form = atoi(e);
        }
        if (wide && form == 1) return launch_conv<3, 4, 2, 2, 4, 2, 2>(P, out_layout, s);
        if (wide && form == 3) return launch_conv<3, 2, 2, 2, 4, 2, 2>(P, out_layout, s);
        if (wide) return launch_conv<3, 8, 2, 2, 4, 2>(P, out_layout, s);
        return launch_conv<3, 8, 4, 1, 2, 1>(P, out_layout, s);
    }
    if (wide) return launch_conv<1, 16, 2, 2, 4, 2>(P, out_layout, s);
    return launch_conv<1, 16, 4, 1, 2, 1>(P, out_layout, s);
}

static bool conv_force_f32() { return conv_strict_f32(); }

extern "C" int mdtile_conv2d_gn_supported(int cout, int cin, int ksize, int flags, int out_layout) {
    if (conv_force_f32() || (flags & MDTILE_CONV_EXACT_F32) || out_layout != 0) return 0;
    return conv_bf16x3_gn_supported(cout, cin, ksize, (flags & MDTILE_CONV_UPSAMPLE2X) ? 1 : 0) ? 1 : 0;
}

extern "C" int mdtile_conv2d_gn(const float* d_x, const float* d_coef, const float* d_w_packed, const float* d_bias, const float* d_residual,
                                float* d_y, int B, int cin, int cout, int H, int W, int ksize, int flags, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_coef && d_w_packed && d_y, "mdtile_conv2d_gn: null argument");
    MDT_CHECK_ARG(B > 0 && B <= 65535 && cin > 0 && cout > 0 && H > 0 && W > 0, "mdtile_conv2d_gn: bad shape B=%d cin=%d cout=%d H=%d W=%d", B, cin, cout, H, W);
    MDT_CHECK_ARG(mdtile_conv2d_gn_supported(cout, cin, ksize, flags, 0),
                  "mdtile_conv2d_gn: no fused pre-activation kernel for cout=%d cin=%d ksize=%d flags=%d (use mdtile_gn_apply + mdtile_conv2d)", cout, cin, ksize, flags);
    MDT_CHECK_ARG((size_t)H * W < (1u << 31), "mdtile_conv2d_gn: input plane of 2^31 or more pixels");
    return conv_bf16x3_launch(d_x, d_w_packed + f32_packed_floats(cout, cin, ksize), d_bias, d_residual, d_y, B, cin, cout, H, W, 0, d_coef,
                              as_stream(stream));
}

// ---- statistics of the output from the conv's own epilogue (slow mode: the input of a POOLED GroupNorm) ---------------------------------
// d_ws = [ mdtile_gn_stats_ws_size(B, groups) bytes of fp64 stage-2 partials | the kernels' fp64 per-block partials ]; sized for the
// finest block grid of the families (8 x 32 px x 128 couts, the fp32 hand-over kernel)
extern "C" size_t mdtile_conv_stats_ws_size(int B, int cout, int H, int W, int groups) {
    if (B <= 0 || cout <= 0 || H <= 0 || W <= 0 || groups <= 0) return 0;
    // the record kernels write per WAVE (4 row groups of a 16 x 32 px item; sub-pixel kernel: 4 row groups x 2 row parities of an 8 x 32 INPUT px item)
    size_t units = (size_t)((W + 31) / 32) * ((H + 7) / 8);
    units = std::max(units, (size_t)conv_rec_stats_units(H, W, 0));
    units = std::max(units, (size_t)conv_rec_stats_units(H + 1, W + 1, 1));
    return mdtile_gn_stats_ws_size(B, groups) + (size_t)B * units * (round_up(cout, 128) / 128) * 64 * sizeof(double);
}

static bool stats_groups_ok(int cout, int groups) {
    return groups > 0 && cout % groups == 0 && (cout / groups) % 4 == 0 && 128 % (cout / groups) == 0;
}

extern "C" int mdtile_conv2d_gn_stats_supported(int cout, int cin, int ksize, int flags, int groups) {
    if (!mdtile_conv2d_gn_supported(cout, cin, ksize, flags, 0)) return 0;
    return conv_bf16x3_stats_supported(cout, cin, ksize, 0) && stats_groups_ok(cout, groups) ? 1 : 0;
}

extern "C" int mdtile_conv2d_gn_stats(const float* d_x, const float* d_coef, const float* d_w_packed, const float* d_bias, const float* d_residual,
                                      float* d_y, int B, int cin, int cout, int H, int W, int ksize, int flags, int groups, float* d_mean,
                                      float* d_var, void* d_ws, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_coef && d_w_packed && d_y && d_mean && d_var && d_ws, "mdtile_conv2d_gn_stats: null argument");
    MDT_CHECK_ARG(B > 0 && B <= 65535 && cin > 0 && cout > 0 && H > 0 && W > 0, "mdtile_conv2d_gn_stats: bad shape B=%d cin=%d cout=%d H=%d W=%d", B, cin, cout, H, W);
    MDT_CHECK_ARG(mdtile_conv2d_gn_stats_supported(cout, cin, ksize, flags, groups),
                  "mdtile_conv2d_gn_stats: no statistics kernel for cout=%d cin=%d ksize=%d flags=%d groups=%d (use mdtile_conv2d_gn + mdtile_gn_stats)", cout, cin, ksize, flags, groups);
    MDT_CHECK_ARG((size_t)H * W < (1u << 31), "mdtile_conv2d_gn_stats: input plane of 2^31 or more pixels");
    double* d_part = reinterpret_cast<double*>(static_cast<char*>(d_ws) + mdtile_gn_stats_ws_size(B, groups));
    hipStream_t s = as_stream(stream);
    const int rc = conv_bf16x3_launch(d_x, d_w_packed + f32_packed_floats(cout, cin, ksize), d_bias, d_residual, d_y, B, cin, cout, H, W, 0, d_coef, s, d_part);
    if (rc != MDTILE_OK) return rc;
    const int units = ((W + 31) / 32) * ((H + 7) / 8);
    return conv_stats_finish_launch(d_part, B, cout, (size_t)H * W, units, cout / 128, 32, groups, d_mean, d_var, d_ws, s);
}

// ---- record-image conv path (vae_conv_rec.hip) ----------------------------------------------------------------------
extern "C" size_t mdtile_rec_size(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || C % 32 != 0 || H <= 0 || W <= 0) return 0;
    return rec_image_bytes(B, C, H, W);
}

static bool rec_image_ok(int B, int C, int H, int W) {
    // per-lane DMA offsets are 32-bit BYTE offsets inside one pair of planes; record indices of a whole image stay below 2^32.
    // The epilogue (conv_rec_common.h) adds 32-bit lane offsets to wave-uniform bases: < 2 record planes, < 5 fp32 planes (HW <= plane records)
    return B > 0 && C > 0 && C % 32 == 0 && H > 0 && W > 0 && H + 2 <= 65535 && (size_t)B * (C / 8) <= 65535 &&
           2 * rec_plane_records(H, W) * 16 < ((size_t)1 << 32);
}

extern "C" int mdtile_rec_from_f32(const float* d_x, const float* d_coef, void* d_rec, int B, int C, int H, int W, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_rec, "mdtile_rec_from_f32: null argument");
    MDT_CHECK_ARG(rec_image_ok(B, C, H, W), "mdtile_rec_from_f32: unsupported shape B=%d C=%d H=%d W=%d (C %% 32 == 0 required)", B, C, H, W);
    return rec_from_f32_launch(d_x, d_coef, d_rec, B, C, H, W, as_stream(stream));
}

extern "C" int mdtile_rec_to_f32(const void* d_rec, float* d_x, int B, int C, int H, int W, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_rec, "mdtile_rec_to_f32: null argument");
    MDT_CHECK_ARG(rec_image_ok(B, C, H, W), "mdtile_rec_to_f32: unsupported shape B=%d C=%d H=%d W=%d", B, C, H, W);
    return rec_to_f32_launch(d_rec, d_x, B, C, H, W, as_stream(stream));
}

extern "C" int mdtile_conv2d_rec_supported(int cout, int cin, int ksize, int flags) {
    if (conv_force_f32() || (flags & MDTILE_CONV_EXACT_F32)) return 0;
    return conv_rec_supported(cout, cin, ksize) ? 1 : 0;
}

// MDTILE_CONV_REC_ONE_BLOCK / _TWO_BLOCKS: the caller names the kernel family (tests, probes); 0 = chosen per launch (rec_two_blocks)
// (bit 16 names the dripped-epilogue probe kernel, in the PROBES twin of the library only: probes/csrc/vae_conv_recd.hip)
static int rec_family(int flags) { return (flags & MDTILE_CONV_REC_ONE_BLOCK) ? 1 : (flags & MDTILE_CONV_REC_TWO_BLOCKS) ? 2 : (kProbes && (flags & 16)) ? 3 : 0; }

extern "C" int mdtile_conv2d_rec(const void* d_x_rec, const float* d_w_packed, const float* d_bias, const float* d_residual, float* d_y,
                                 void* d_y_rec, const float* d_y_coef, int B, int cin, int cout, int H, int W, int flags,
                                 mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x_rec && d_w_packed && (d_y || d_y_rec), "mdtile_conv2d_rec: null argument (one of d_y / d_y_rec is required)");
    MDT_CHECK_ARG(mdtile_conv2d_rec_supported(cout, cin, 3, flags), "mdtile_conv2d_rec: no record kernel for cout=%d cin=%d flags=%d", cout, cin, flags);
    const int up = (flags & MDTILE_CONV_UPSAMPLE2X) ? 1 : 0;
    MDT_CHECK_ARG(!up || (H % 2 == 0 && W % 2 == 0), "mdtile_conv2d_rec: upsample2x needs even output size, got %dx%d", H, W);
    MDT_CHECK_ARG(rec_image_ok(B, cin, up ? H / 2 : H, up ? W / 2 : W) && (cout % 32 != 0 || rec_image_ok(B, cout, H, W)),
                  "mdtile_conv2d_rec: unsupported shape B=%d cin=%d cout=%d H=%d W=%d", B, cin, cout, H, W);
    MDT_CHECK_ARG(cout % 128 == 0 || (!up && !d_y_rec && !d_residual),
                  "mdtile_conv2d_rec: the narrow (cout < 32) kernel writes fp32 only, no residual, no upsample (cout=%d)", cout);
    return conv_rec_launch(d_x_rec, d_w_packed + f32_packed_floats(cout, cin, 3), d_bias, d_residual, d_y, d_y_rec, d_y_coef, B, cin, cout,
                           H, W, up, as_stream(stream), nullptr, rec_family(flags));
}

extern "C" int mdtile_conv2d_rec_stats_supported(int cout, int cin, int ksize, int flags, int groups) {
    return mdtile_conv2d_rec_supported(cout, cin, ksize, flags) && cout % 128 == 0 && stats_groups_ok(cout, groups) ? 1 : 0;
}

extern "C" int mdtile_conv2d_rec_stats(const void* d_x_rec, const float* d_w_packed, const float* d_bias, const float* d_residual, float* d_y,
                                       int B, int cin, int cout, int H, int W, int flags, int groups, float* d_mean, float* d_var, void* d_ws,
                                       mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x_rec && d_w_packed && d_y && d_mean && d_var && d_ws, "mdtile_conv2d_rec_stats: null argument");
    MDT_CHECK_ARG(mdtile_conv2d_rec_stats_supported(cout, cin, 3, flags, groups),
                  "mdtile_conv2d_rec_stats: no statistics kernel for cout=%d cin=%d flags=%d groups=%d", cout, cin, flags, groups);
    const int up = (flags & MDTILE_CONV_UPSAMPLE2X) ? 1 : 0;
    MDT_CHECK_ARG(!up || (H % 2 == 0 && W % 2 == 0), "mdtile_conv2d_rec_stats: upsample2x needs even output size, got %dx%d", H, W);
    MDT_CHECK_ARG(rec_image_ok(B, cin, up ? H / 2 : H, up ? W / 2 : W) && rec_image_ok(B, cout, H, W),
                  "mdtile_conv2d_rec_stats: unsupported shape B=%d cin=%d cout=%d H=%d W=%d", B, cin, cout, H, W);
    hipStream_t s = as_stream(stream);
    const float* w = d_w_packed + f32_packed_floats(cout, cin, 3);
    if (rec_family(flags) == 0 && !conv_rec_stats_in_epilogue(B, cin, cout, H, W, up)) {
        // a launch of a few item rounds: the two-blocks-per-CU family gains more than the statistics pass costs (conv_rec_stats_in_epilogue)
        const int rc = conv_rec_launch(d_x_rec, w, d_bias, d_residual, d_y, nullptr, nullptr, B, cin, cout, H, W, up, s);
        return rc != MDTILE_OK ? rc : mdtile_gn_stats(d_y, B, cout, H * W, groups, d_mean, d_var, d_ws, stream);
    }
    MDT_CHECK_ARG(rec_family(flags) <= 1, "mdtile_conv2d_rec_stats: only the one-block-per-CU family leaves statistics (flags=%d)", flags);
    double* d_part = reinterpret_cast<double*>(static_cast<char*>(d_ws) + mdtile_gn_stats_ws_size(B, groups));
    const int rc = conv_rec_launch(d_x_rec, w, d_bias, d_residual, d_y, nullptr, nullptr, B, cin, cout, H, W, up, s, nullptr, 1, d_part);
    if (rc != MDTILE_OK) return rc;
    return conv_stats_finish_launch(d_part, B, cout, (size_t)H * W, conv_rec_stats_units(H, W, up), cout / 128, 32, groups, d_mean, d_var, d_ws, s);
}

// Nearest-2x + 3x3 conv of a WINDOW of the input record image (live-window narrowing of the decoder tiles, see include/mdtile.h)
extern "C" int mdtile_upconv2d_rec_window(const void* d_x_rec, const float* d_w_packed, const float* d_bias, float* d_y, void* d_y_rec,
                                          const float* d_y_coef, int B, int cin, int cout, int Hin, int Win, const int* y0, const int* x0,
                                          int h, int w, int flags, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x_rec && d_w_packed && (d_y || d_y_rec) && y0 && x0, "mdtile_upconv2d_rec_window: null argument (one of d_y / d_y_rec is required)");
    MDT_CHECK_ARG(mdtile_conv2d_rec_supported(cout, cin, 3, 0) && cout % 128 == 0, "mdtile_upconv2d_rec_window: no record kernel for cout=%d cin=%d", cout, cin);
    MDT_CHECK_ARG(B >= 1, "mdtile_upconv2d_rec_window: B=%d", B);
    int win[2 + 2 * 8] = {Hin, Win};
    for (int b = 0; b < B; ++b) {
        MDT_CHECK_ARG(h >= 1 && w >= 1 && y0[b] >= 0 && x0[b] >= 0 && y0[b] + h <= Hin && x0[b] + w <= Win,
                      "mdtile_upconv2d_rec_window: window y0=%d x0=%d h=%d w=%d of image %d leaves the %dx%d input", y0[b], x0[b], h, w, b, Hin, Win);
        // the kernel keeps 8 origins (image b uses slot b & 7): more images only when they repeat with that period
        MDT_CHECK_ARG(b < 8 || (y0[b] == y0[b & 7] && x0[b] == x0[b & 7]),
                      "mdtile_upconv2d_rec_window: more than 8 images need window origins that repeat every 8 images (image %d)", b);
    }
    for (int b = 0; b < 8; ++b) {
        win[2 + 2 * b] = y0[b % B];
        win[3 + 2 * b] = x0[b % B];
    }
    MDT_CHECK_ARG(rec_image_ok(B, cin, Hin, Win) && rec_image_ok(B, cout, 2 * h, 2 * w),
                  "mdtile_upconv2d_rec_window: unsupported shape B=%d cin=%d cout=%d Hin=%d Win=%d", B, cin, cout, Hin, Win);
    return conv_rec_launch(d_x_rec, d_w_packed + f32_packed_floats(cout, cin, 3), d_bias, nullptr, d_y, d_y_rec, d_y_coef, B, cin, cout,
                           2 * h, 2 * w, 1, as_stream(stream), win, rec_family(flags));
}

// ldm Downsample: y = conv3x3_stride2(pad(x, right 1, bottom 1)); output (Hin - 2) / 2 + 1 rows (likewise columns).
// Weights: the image of mdtile_conv_pack(ksize 3).  Split-bf16 stride-2 kernel where the 3x3 family is eligible (round 3: the exact
// kernel made the three downsample convs 9 % of an 8K encode); exact-fp32 MFMA kernel otherwise / under MDTILE_PRECISION_F32.
extern "C" int mdtile_conv2d_down2(const float* d_x, const float* d_w_packed, const float* d_bias, float* d_y, int B, int cin, int cout,
                                   int Hin, int Win, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_x && d_w_packed && d_y, "mdtile_conv2d_down2: null argument");
    MDT_CHECK_ARG(B > 0 && B <= 65535 && cin > 0 && cout > 0 && Hin >= 2 && Win >= 2, "mdtile_conv2d_down2: bad shape B=%d cin=%d cout=%d Hin=%d Win=%d",
                  B, cin, cout, Hin, Win);
    MDT_CHECK_ARG((size_t)Hin * Win < (1u << 31), "mdtile_conv2d_down2: input plane of 2^31 or more pixels");
    if (!conv_strict_f32() && conv_bf16x3_eligible(cout, cin, 3))      // split-bf16 stride-2 kernel (vae_conv_bf16x3.hip, S = 2)
        return conv_bf16x3_down2_launch(d_x, d_w_packed + f32_packed_floats(cout, cin, 3), d_bias, d_y, B, cin, cout, Hin, Win, as_stream(stream));
    ConvParams P;
    P.x = d_x; P.w = d_w_packed; P.bias = d_bias; P.res = nullptr; P.y = d_y;
    P.B = B; P.Cin = cin; P.Cout = cout; P.CoutP = round_up(cout, 32);
    P.H = (Hin - 2) / 2 + 1; P.W = (Win - 2) / 2 + 1;
    P.Hin = Hin; P.Win = Win; P.up = 0;
    hipStream_t s = as_stream(stream);
    if (P.CoutP > 64) return launch_conv_down2<8, 2, 2, 4, 2>(P, s);
    return launch_conv_down2<8, 4, 1, 2, 1>(P, s);
}
