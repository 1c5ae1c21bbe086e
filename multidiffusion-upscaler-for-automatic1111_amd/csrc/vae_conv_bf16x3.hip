// 3x3 conv tiles of the Tiled-VAE task queue on the bf16 matrix cores with SPLIT-fp32 operands ("bf16x3").
//
// Why: the decoder's 3x3 convs are ~70 % of an 8K decode and gfx950 has no TF32; exact-fp32 MFMA peaks at 157 TFLOP/s,
// bf16 MFMA at 2.5 PFLOP/s.  Every fp32 operand is split x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significand
// bits together) and a product is formed as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  with fp32 accumulation inside the MFMA:
// three bf16 MFMAs per fp32-class product = 16/3 x the fp32-MFMA rate.  The dropped a_lo*b_lo term and the split residue
// are <= 2^-16 relative per product (measured end-to-end against the fp32 reference in tests/test_gpu_vae.py: ~1e-5, the
// stated tolerance is 1e-3).  The exact-fp32 kernel (vae_conv.hip) stays available (MDTILE_CONV_EXACT_F32).
//
// Upstream call sites replaced: the nn.Conv2d tasks conv1 / conv2 / upsample.conv of scripts/tilevae.py:115-195 (+ the
// queue's add_res, tilevae.py:614-616, and F.interpolate(nearest, 2x) of ldm's Upsample, both fused as in vae_conv.hip).
//
// GEMM view (implicit, no im2col):  D[cout][px] = sum_{tap, cin} W[cout][cin][tap] * X[cin][px + tap]
//   MFMA v_mfma_f32_32x32x16_bf16:  A = weights (M = 32 couts, K = 16 cin of one tap), B = input (K = 16 cin, N = 32 px of a row)
//   lane l supplies 8 consecutive k of row/col (l & 31): k-half (l >> 5)  ->  both operands want "8 channels of one
//   cout / one pixel" as one 16-byte LDS record:
//     input  LDS image  [hl][cg = 2][rows = 10][cols = 34] x 16 B   (cg = 8-channel group; halo tile of an 8 x 32 px block)
//     weight LDS image  [hl][dx = 3][mtile][lane = 64]     x 16 B   (exactly the global pre-packed order: straight copy)
//   every fragment read is one conflict-free ds_read_b128 (32 consecutive 16-byte records per half-wave).
// Block = 512 threads = 8 waves (2 per SIMD), output tile BM couts x 8 rows x 32 px; wave = 64 couts x NROW rows.
// K loop = phases (16-channel K-step, dy): 3 taps x (2 x NROW) tiles x 3 MFMAs per wave and phase, ONE barrier per phase;
// the next phase's weights (24 KB, L2-resident) and the next K-step's input slab are fetched into registers at the start
// of a phase and written to the other LDS stage at its end (fp32 -> hi/lo split happens on that write).
#include "common.h"

using namespace mdt;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // one 16-byte record (plain vector type: stays in registers)

namespace {

struct ConvBParams {
    const float* x;      // [B, Cin, Hin, Win] fp32
    const u32x4* w;      // packed bf16 hi/lo records, see k_conv_pack_bf16x3
    const float* bias;   // [Cout] or null
    const float* res;    // residual [B, Cout, H, W] or null
    float* y;            // [B, Cout, H, W]
    int B, Cin, Cout, H, W;   // H, W: OUTPUT spatial size
    int Hin, Win, up;         // input spatial size; up = 1: sub-pixel upsample kernel (output = 2x input)
    int ptiles, PX, NCB, NK;  // pixel tiles, tiles per row, cout blocks, 16-channel K-steps
    const float* coef;        // fused pre-activation (GNS kernels): [B][2][Cin] = per-channel scale a, shift s; x' = silu(a x + s)
    int perm;                 // channel order inside a K-step, see kstep_c0 / kstep_cj
    double* gn_part;          // ST kernels: per-block partial (sum, sum of squares) of the OUTPUT per 4-cout quad, see the epilogue
};

constexpr int MAX_GN_CIN = 512;   // the fused pre-activation keeps a[Cin], s[Cin] in LDS

constexpr int TH = 8, TW = 32, ROWS = TH + 2, COLS = TW + 2;
constexpr int IN_REC = 2 * ROWS * COLS;                 // 680 records (16 B) per hl per stage

__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = (__bf16)v[i];                            // v_cvt_pk_bf16_f32 (RNE)
        l[i] = (__bf16)(v[i] - (float)h[i]);            // exact residue, rounded once
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

// Channel order inside a 16-channel K-step.  perm = 1 (cin % 32 == 0): K index (kg, j) of K-step k is channel
//     32*(k>>1) + 16*(k&1) + 4*kg + (j&3) + 8*(j>>2)
// = the channels one lane of a 32x32 MFMA accumulator tile owns, which is the order the record-image conv family
// (vae_conv_rec.hip) stores activations in; ONE packed weight image then serves both families.  perm = 0: k*16 + kg*8 + j.
__device__ __host__ __forceinline__ int kstep_c0(int k, int kg, int perm) { return perm ? 32 * (k >> 1) + 16 * (k & 1) + 4 * kg : k * 16 + kg * 8; }
__device__ __host__ __forceinline__ int kstep_cj(int j, int perm) { return perm ? (j & 3) + 8 * (j >> 2) : j; }

// GNS = true: the fixed-statistics GroupNorm + SiLU that precedes conv1 / conv2 in every resblock (custom_group_norm +
// inplace_nonlinearity, scripts/tilevae.py:218-245, 102-104) is applied to the input while it is staged:
// x' = silu(fma(x, a[c], s[c])) with a = gamma * rstd, s = beta - mean * a (mdtile_gn_coeffs) -- the normalised activation is
// never written to HBM (1R + 1W of every pre-conv activation saved).  Zero padding applies to x' (mask after the transform).
// Two block shapes (every other variant was measured and dropped):
//   MT = 4 (128 couts): weights by LDS-DMA (already in fragment order), ONE input stage (an extra barrier before it is
//           overwritten, once per K-step), 124 VGPR + 71 KB LDS -> TWO blocks per CU (4 waves per SIMD); MFMAs issued
//           term-major across the accumulators of a tap column (a dependent MFMA is >= 4 MFMAs away)
//   MT = 2 (64 couts, small decoders): weights through VGPRs, two input stages
// S = 2 (round 3): ldm's Downsample of the ENCODER (scripts/tilevae.py:155-171: conv 3x3, stride 2, over pad(x, right 1, bottom 1)) on
// the same split-bf16 arithmetic.  Output tile 8 x 32 px, input halo tile 17 x 65; the LDS image keeps the two COLUMN PARITIES of a
// row apart ([cg][row 17][parity 2][33]) so that the fragment of tap dx -- input columns 2 x + dx of 32 consecutive output px -- is
// again 32 consecutive records (a stride-2 ds_read_b128 would be a 2-way bank conflict).  Zero padding only right / bottom.
// ST = true (slow mode, round 5): the conv whose output feeds a POOLED GroupNorm (GroupNormParam.add_tile -> get_var_mean,
// scripts/tilevae.py:207-215, 300-307) also leaves the statistics of that output: every block writes (sum, sum of squares) of its
// BM couts x 8 x 32 px in 4-cout quads -- gn_part[(b * ptiles + ptile) * NCB + cb][BM / 4][2] fp64, combined in a fixed order by
// k_conv_stats_partial / k_gn_final (vae_norm.hip): the separate pass that re-read the whole activation is gone.  y is bit-identical.
template <int MT, bool GNS, int S = 1, bool ST = false>
__global__ __launch_bounds__(512, (MT == 4 && S == 1) ? 4 : 2) void k_conv3x3_bf16x3(const ConvBParams P) {
    constexpr int TH_ = 8;
    constexpr bool WDMA = MT == 4, IB1 = MT == 4 || S == 2, TERM_MAJOR = MT == 4;   // (S = 2: the 72 KB halo tile exists once)
    constexpr int BM = MT * 32;
    constexpr int WAVES_M = MT / 2, WAVES_R = 8 / WAVES_M, NROW = TH_ / WAVES_R;
    constexpr int HCOL = TW + 1;                                   // S = 2: records per column parity of a row
    constexpr int COLSL = S == 1 ? COLS : 2 * HCOL;                // records per LDS row
    constexpr int ROWS_ = S == 1 ? TH_ + 2 : 2 * TH_ + 1, IN_REC_ = 2 * ROWS_ * COLSL;   // halo tile records per hl per stage
    constexpr int NPASS = (IN_REC_ + 511) / 512;                  // staging passes of the 512 threads
    constexpr int W_REC = 2 * 3 * MT * 64;              // records per weight chunk (hi block then lo block)
    constexpr int NWREG = (W_REC + 511) / 512;          // 3 (MT = 4) or 2 (MT = 2, half of the threads on the 2nd)
    // LDS (16-byte records): input [2 stages][hl][IN_REC_], weights [2 stages][W_REC]
    constexpr int IN_STAGE = 2 * IN_REC_, NIST = IB1 ? 1 : 2, WST = 2;
    __shared__ u32x4 smem[NIST * IN_STAGE + WST * W_REC];
    __shared__ float4 coef_l[GNS ? 2 * MAX_GN_CIN / 4 : 1];   // a[0..Cin) at 0, s[0..Cin) at MAX_GN_CIN
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + NIST * IN_STAGE;

    // ---- block -> (pixel tile, cout block): XCD = id % 8 keeps all cout blocks of a pixel tile on one L2
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int ptile = (slot / P.NCB) * 8 + xcd, cb = slot % P.NCB;
    if (ptile >= P.ptiles) return;
    const int b = blockIdx.y;
    const int py = ptile / P.PX, px = ptile - py * P.PX;
    const int y0 = py * TH_, x0 = px * TW;
    if (GNS) {
        float* cl = reinterpret_cast<float*>(coef_l);
        const float* cg = P.coef + (size_t)b * 2 * P.Cin;
        for (int c = threadIdx.x; c < P.Cin; c += 512) {
            cl[c] = cg[c];
            cl[MAX_GN_CIN + c] = cg[P.Cin + c];
        }
        __syncthreads();
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kg = lane >> 5;
    const int wm = wave % WAVES_M, wr = wave / WAVES_M;
    const size_t HWin = (size_t)P.Hin * P.Win;
    const float* xb = P.x + (size_t)b * P.Cin * HWin;

    // ---- input staging map: record s = tid + 512 i -> (cg, r, c); source offset inside a channel plane, valid flag
    // (zero padding).  Loads are unconditional at a clamped address and zeroed by the flag: straight-line code.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // provably wave-uniform
    int soff[NPASS], scg[NPASS];
    float smask[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        int s = tid + 512 * i;
        if (s >= IN_REC_) s = IN_REC_ - 1;                      // lanes past the end shadow the last record (never stored)
        const int cg = s / (ROWS_ * COLSL), p = s - cg * (ROWS_ * COLSL);
        const int r = p / COLSL, c = p - r * COLSL;
        int gy, gx;
        bool inside;
        if (S == 1) {
            gy = y0 + r - 1; gx = x0 + c - 1;
            inside = gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
        } else {           // LDS column c = parity * HCOL + i  <->  input column 2 i + parity of the halo tile (65 of the 66 slots are used)
            const int par = c / HCOL, ci = c - par * HCOL, col = 2 * ci + par;
            gy = 2 * y0 + r; gx = 2 * x0 + col;
            inside = col <= 2 * TW && gy < P.Hin && gx < P.Win;
        }
        scg[i] = cg;
        soff[i] = inside ? gy * P.Win + gx : 0;
        smask[i] = inside ? 1.0f : 0.0f;
    }
    float rin[NPASS][8];
    u32x4 rwt[WDMA ? 1 : NWREG];

    auto load_input = [&](int k) {       // K-step k: channels 16k .. 16k+15
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            if (wave_u * 64 + 512 * i < IN_REC_) {               // whole waves past the end of the record list skip the pass
                const float* src = xb + (size_t)kstep_c0(k, scg[i], P.perm) * HWin + soff[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) rin[i][j] = src[(size_t)kstep_cj(j, P.perm) * HWin];
            }
        }
    };
    auto store_input = [&](int stage, int k) {   // k: the K-step held in rin (channels 16k ..)
        u32x4* dst = in_l + stage * IN_STAGE;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            if (wave_u * 64 + 512 * i < IN_REC_) {
                float v[8];
                if (GNS) {
                    const int c4 = kstep_c0(k, scg[i], P.perm) >> 2, c4b = c4 + (P.perm ? 2 : 1);   // channels j = 4..7 sit 8 (perm) or 4 further
                    const float4 a0 = coef_l[c4], a1 = coef_l[c4b], s0 = coef_l[MAX_GN_CIN / 4 + c4], s1 = coef_l[MAX_GN_CIN / 4 + c4b];
                    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float t = fmaf(rin[i][j], av[j], sv[j]);
                        // silu(t) = t / (1 + e^-t): v_exp_f32 + v_rcp_f32 (<= ~2 ulp; the operands are rounded to 16 bits next)
                        v[j] = smask[i] != 0.0f ? t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)) : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = rin[i][j] * smask[i];
                }
                u32x4 hi, lo;
                split8(v, hi, lo);
                if (tid + 512 * i < IN_REC_) {
                    dst[tid + 512 * i] = hi;
                    dst[IN_REC_ + tid + 512 * i] = lo;
                }
            }
        }
    };
    const u32x4* wsrc = P.w + (size_t)cb * P.NK * 3 * W_REC;
    auto load_weights = [&](int ph) {    // phase ph = k*3 + dy: one contiguous chunk of W_REC records (-> LDS stage ph & 1)
        const u32x4* src = wsrc + (size_t)ph * W_REC;
#pragma unroll
        for (int i = 0; i < NWREG; ++i)
            if (wave_u * 64 + 512 * i < W_REC) {   // W_REC is a multiple of 64: whole waves
                if (WDMA)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + tid + 512 * i),
                                                     (__attribute__((address_space(3))) void*)(w_l + (ph % WST) * W_REC + wave_u * 64 + 512 * i), 16, 0, 0);
                else
                    rwt[WDMA ? 0 : i] = src[tid + 512 * i];
            }
    };
    auto store_weights = [&](int stage) {
        if (WDMA) {
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's part of the chunk has landed (the barrier covers the others)
            return;
        }
        u32x4* dst = w_l + stage * W_REC;
#pragma unroll
        for (int i = 0; i < NWREG; ++i)
            if (wave_u * 64 + 512 * i < W_REC) dst[tid + 512 * i] = rwt[WDMA ? 0 : i];
    };

    f32x16 acc[2][NROW];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NROW; ++n)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.0f;

    const int nph = P.NK * 3;
    load_input(0);
    load_weights(0);
    store_input(0, 0);
    store_weights(0);
    __syncthreads();

    for (int ph = 0; ph < nph; ++ph) {
        const int k = ph / 3, dy = ph - 3 * k;
        if (dy == 0 && k + 1 < P.NK) load_input(k + 1);
        if (ph + 1 < nph) load_weights(ph + 1);

        const u32x4* wst = w_l + (ph % WST) * W_REC;
        const u32x4* ist = in_l + (IB1 ? 0 : (k & 1)) * IN_STAGE;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            bf16x8 a[2][2];   // [m][hl]
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    a[m][hl] = __builtin_bit_cast(bf16x8, wst[((hl * 3 + dx) * MT + wm * 2 + m) * 64 + lane]);
                }
            if (TERM_MAJOR) {
                bf16x8 bh[NROW], bl[NROW];
#pragma unroll
                for (int n = 0; n < NROW; ++n) {
                    const int rec = S == 1 ? (kg * ROWS_ + wr * NROW + n + dy) * COLS + l31 + dx
                                           : (kg * ROWS_ + 2 * (wr * NROW + n) + dy) * COLSL + (dx & 1) * HCOL + l31 + (dx >> 1);
                    bh[n] = __builtin_bit_cast(bf16x8, ist[rec]);
                    bl[n] = __builtin_bit_cast(bf16x8, ist[IN_REC_ + rec]);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
#pragma unroll
                    for (int n = 0; n < NROW; ++n)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][t == 0 ? 1 : 0], t == 1 ? bl[n] : bh[n], acc[m][n], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0x07F7);   // everything but MFMAs may cross: keeps the term-major order
                }
            } else {
#pragma unroll
                for (int n = 0; n < NROW; ++n) {
                    const int rec = S == 1 ? (kg * ROWS_ + wr * NROW + n + dy) * COLS + l31 + dx
                                           : (kg * ROWS_ + 2 * (wr * NROW + n) + dy) * COLSL + (dx & 1) * HCOL + l31 + (dx >> 1);
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, ist[rec]), bl = __builtin_bit_cast(bf16x8, ist[IN_REC_ + rec]);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], bh, acc[m][n], 0, 0, 0);   // w_lo * x_hi
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], bl, acc[m][n], 0, 0, 0);   // w_hi * x_lo
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], bh, acc[m][n], 0, 0, 0);   // w_hi * x_hi
                    }
                }
            }
        }

        if (ph + 1 < nph) store_weights((ph + 1) & 1);
        if (dy == 2 && k + 1 < P.NK) {
            if (IB1) __syncthreads();      // single input stage: every wave must be done reading K-step k before it is overwritten
            store_input(IB1 ? 0 : ((k + 1) & 1), k + 1);
        }
        __syncthreads();
    }

    // ---- epilogue: + bias (+ residual), store NCHW.  C/D layout of a 32x32 MFMA: col = lane & 31, row = (q&3) + 8*(q>>2) + 4*(lane>>5)
    // (128-byte runs along x per register).  All loads of a tile are issued before the first store.
    const size_t HW = (size_t)P.H * P.W;
    const int x = x0 + l31;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int cbase = cb * BM + (wm * 2 + m) * 32 + 4 * kg;
        float bq[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = cbase + (q & 3) + 8 * (q >> 2);
            bq[q] = P.bias ? P.bias[co < P.Cout ? co : P.Cout - 1] : 0.0f;
        }
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};   // ST: this lane's part of the four quads of the tile (rows 8 j + 4 kg ..)
#pragma unroll
        for (int n = 0; n < NROW; ++n) {
            const int y = y0 + wr * NROW + n;
            if (y < P.H && x < P.W) {
                const size_t o0 = ((size_t)b * P.Cout) * HW + (size_t)y * P.W + x;
                float rq[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = cbase + (q & 3) + 8 * (q >> 2);
                    rq[q] = P.res ? P.res[o0 + (size_t)(co < P.Cout ? co : P.Cout - 1) * HW] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = cbase + (q & 3) + 8 * (q >> 2);
                    const float v = acc[m][n][q] + bq[q] + rq[q];
                    if (co < P.Cout) P.y[o0 + (size_t)co * HW] = v;
                    if (ST) {                  // (the launcher only takes Cout % BM == 0 here: every cout of the block exists)
                        s1[q >> 2] += v;
                        s2[q >> 2] = fmaf(v, v, s2[q >> 2]);
                    }
                }
            }
        }
        if (ST) {
            // a lane's fp32 sums cover 2 rows x 4 couts; from here on fp64 (an fp32 tree over the block would round at 1e-7 of the BLOCK's sum
            // of squares, which var = E[x^2] - mean^2 amplifies by mean^2 / var).  The 32 pixels of a row sit in the 32 lanes of a
            // half-wave: xor offsets < 32 stay inside it.
            double d1[4], d2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d1[j] = (double)s1[j];
                d2[j] = (double)s2[j];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) {
                    d1[j] += __shfl_xor(d1[j], off, 64);
                    d2[j] += __shfl_xor(d2[j], off, 64);
                }
            }
            if (l31 == 0) {
                double2* sl = reinterpret_cast<double2*>(smem);      // the K loop ended behind a barrier: the operand stages are free
#pragma unroll
                for (int j = 0; j < 4; ++j) sl[wr * (BM / 4) + (wm * 2 + m) * 8 + 2 * j + kg] = make_double2(d1[j], d2[j]);
            }
        }
    }
    if (ST) {
        __syncthreads();
        if (tid < BM / 2) {            // (quad, sum | sum of squares): the row groups of the block in a fixed order
            const double* sl = reinterpret_cast<const double*>(smem);
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < WAVES_R; ++r) t += sl[r * (BM / 2) + tid];
            P.gn_part[(((size_t)b * P.ptiles + ptile) * P.NCB + cb) * (BM / 2) + tid] = t;
        }
    }
}

// OIHW fp32 -> records [cb][k][dy][hl][dx][mt][lane] of 8 bf16:  cout = cb*BM + mt*32 + (lane & 31),
// cin = k*16 + (lane >> 5)*8 + j,  tap = (dy, dx);  hl = 0: bf16(w), hl = 1: bf16(w - hi).  Zero outside [Cout) x [Cin).
__global__ void k_conv_pack_bf16x3(const float* __restrict__ w, u32x4* __restrict__ out, int Cout, int Cin, int MT, int NCB, int NK, int perm) {
    const size_t n = (size_t)NCB * NK * 3 * 2 * 3 * MT * 64;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t r = i;
    const int lane = (int)(r % 64); r /= 64;
    const int mt = (int)(r % MT); r /= MT;
    const int dx = (int)(r % 3); r /= 3;
    const int hl = (int)(r % 2); r /= 2;
    const int dy = (int)(r % 3); r /= 3;
    const int k = (int)(r % NK); r /= NK;
    const int cb = (int)r;
    const int co = cb * MT * 32 + mt * 32 + (lane & 31);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = kstep_c0(k, lane >> 5, perm) + kstep_cj(j, perm);
        const float v = (co < Cout && ci < Cin) ? w[(((size_t)co * Cin + ci) * 3 + dy) * 3 + dx] : 0.0f;
        const __bf16 h = (__bf16)v;
        o[j] = hl == 0 ? h : (__bf16)(v - (float)h);
    }
    out[i] = __builtin_bit_cast(u32x4, o);
}


// ---------------------------------------------------------------------------------------------------------------------
// Fused nearest-2x upsample + 3x3 conv as FOUR 2x2 convs on the un-upsampled input ("sub-pixel" form): 2.25x fewer MACs.
//
// ldm's Upsample is F.interpolate(x, 2.0, 'nearest') followed by a 3x3 'same' conv (the queue's upsample.conv task,
// scripts/tilevae.py:139-153).  up[i][j] = in[i >> 1][j >> 1], so for output pixel (2y + a, 2x + b), a, b in {0, 1}, the
// three up-rows 2y+a-1 .. 2y+a+1 land on only TWO input rows:
//     a = 0:  rows {y-1, y, y}   ->  tap u=0 = w[dy=0] on row y-1,        tap u=1 = w[dy=1] + w[dy=2] on row y
//     a = 1:  rows {y, y, y+1}   ->  tap u=0 = w[dy=0] + w[dy=1] on row y, tap u=1 = w[dy=2] on row y+1
// i.e. input row offset = a + u - 1; columns alike with (b, v).  Zero padding is preserved exactly: up-row -1 <-> input
// row -1 and up-row 2H <-> input row H are the only out-of-range taps and the merged taps never straddle the border.
// The merged weights are summed in fp32 once at pack time (k_upconv_pack_bf16x3), then hi/lo split like the others.
//
// Block = 512 threads, BM couts x (8 x 32 INPUT px) = 16 x 64 output px of ONE row parity a (blockIdx carries a) and BOTH
// column parities: the accumulators of b = 0 and b = 1 sit in the same lane, so the epilogue stores float2 (x = 2X, 2X+1).
// K loop = phases (16-channel K-step, u): column shifts s = b + v in {0, 1, 2} share their input fragments between the
// parities -> per phase and wave 3 x NROW x 2 input + 4 x 2 x 2 weight fragment reads feed 4 x 2 x NROW x 3 MFMAs.
template <int MT>
__global__ __launch_bounds__(512) void k_upconv_bf16x3(const ConvBParams P) {
    constexpr int BM = MT * 32;
    constexpr int WAVES_M = MT / 2, WAVES_R = 8 / WAVES_M, NROW = TH / WAVES_R;
    constexpr int W_REC = 2 * 2 * 2 * MT * 64;          // [hl][b][v][mt][lane] records per (a, cb, k, u) chunk
    constexpr int NWREG = W_REC / 512;                   // 4 (MT = 4) or 2 (MT = 2)
    constexpr int IN_STAGE = 2 * IN_REC;
    __shared__ u32x4 smem[2 * IN_STAGE + 2 * W_REC];
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + 2 * IN_STAGE;

    // block -> (input pixel tile, cout block, row parity); XCD = id % 8 keeps every (cb, a) of a pixel tile on one L2
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int per = P.NCB * 2;
    const int ptile = (slot / per) * 8 + xcd, rem = slot % per, cb = rem >> 1, a = rem & 1;
    if (ptile >= P.ptiles) return;
    const int b = blockIdx.y;
    const int py = ptile / P.PX, px = ptile - py * P.PX;
    const int y0 = py * TH, x0 = px * TW;              // INPUT coordinates

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kg = lane >> 5;
    const int wm = wave % WAVES_M, wr = wave / WAVES_M;
    const size_t HWin = (size_t)P.Hin * P.Win;
    const float* xb = P.x + (size_t)b * P.Cin * HWin;

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool has_rec1 = wave_u * 64 + 512 < IN_REC;
    int soff[2], scg[2];
    float smask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int s = tid + 512 * i;
        if (s >= IN_REC) s = IN_REC - 1;
        const int cg = s / (ROWS * COLS), p = s - cg * (ROWS * COLS);
        const int r = p / COLS, c = p - r * COLS;
        const int gy = y0 + r - 1, gx = x0 + c - 1;
        const bool inside = gy >= 0 && gy < P.Hin && gx >= 0 && gx < P.Win;
        scg[i] = cg;
        soff[i] = inside ? gy * P.Win + gx : 0;
        smask[i] = inside ? 1.0f : 0.0f;
    }
    float rin[2][8];
    u32x4 rwt[NWREG];

    auto load_input = [&](int k) {
        {
            const float* src = xb + (size_t)kstep_c0(k, scg[0], P.perm) * HWin + soff[0];
#pragma unroll
            for (int j = 0; j < 8; ++j) rin[0][j] = src[(size_t)kstep_cj(j, P.perm) * HWin];
        }
        if (has_rec1) {
            const float* src = xb + (size_t)kstep_c0(k, scg[1], P.perm) * HWin + soff[1];
#pragma unroll
            for (int j = 0; j < 8; ++j) rin[1][j] = src[(size_t)kstep_cj(j, P.perm) * HWin];
        }
    };
    auto store_input = [&](int stage) {
        u32x4* dst = in_l + stage * IN_STAGE;
        {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = rin[0][j] * smask[0];
            u32x4 hi, lo;
            split8(v, hi, lo);
            dst[tid] = hi;
            dst[IN_REC + tid] = lo;
        }
        if (has_rec1) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = rin[1][j] * smask[1];
            u32x4 hi, lo;
            split8(v, hi, lo);
            if (tid + 512 < IN_REC) {
                dst[tid + 512] = hi;
                dst[IN_REC + tid + 512] = lo;
            }
        }
    };
    const int nph = P.NK * 2;
    const u32x4* wsrc = P.w + ((size_t)a * P.NCB + cb) * nph * W_REC;
    auto load_weights = [&](int ph) {
        const u32x4* src = wsrc + (size_t)ph * W_REC;
#pragma unroll
        for (int i = 0; i < NWREG; ++i) rwt[i] = src[tid + 512 * i];
    };
    auto store_weights = [&](int stage) {
        u32x4* dst = w_l + stage * W_REC;
#pragma unroll
        for (int i = 0; i < NWREG; ++i) dst[tid + 512 * i] = rwt[i];
    };

    f32x16 acc[2][2][NROW];   // [b][m][n]
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NROW; ++n)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[bb][m][n][q] = 0.0f;

    load_input(0);
    load_weights(0);
    store_input(0);
    store_weights(0);
    __syncthreads();

    for (int ph = 0; ph < nph; ++ph) {
        const int k = ph >> 1, u = ph & 1;
        if (u == 0 && k + 1 < P.NK) load_input(k + 1);
        if (ph + 1 < nph) load_weights(ph + 1);

        const u32x4* wst = w_l + (ph & 1) * W_REC;
        const u32x4* ist = in_l + (k & 1) * IN_STAGE;
        const int rbase = kg * ROWS + wr * NROW + a + u;   // halo row of output row n: + n
#pragma unroll
        for (int s = 0; s < 3; ++s) {                      // column shift s = b + v
            bf16x8 bh[NROW], bl[NROW];
#pragma unroll
            for (int n = 0; n < NROW; ++n) {
                const int rec = (rbase + n) * COLS + l31 + s;
                bh[n] = __builtin_bit_cast(bf16x8, ist[rec]);
                bl[n] = __builtin_bit_cast(bf16x8, ist[IN_REC + rec]);
            }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int v = s - bb;
                if (v < 0 || v > 1) continue;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, wst[(((0 * 2 + bb) * 2 + v) * MT + wm * 2 + m) * 64 + lane]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, wst[(((1 * 2 + bb) * 2 + v) * MT + wm * 2 + m) * 64 + lane]);
#pragma unroll
                    for (int n = 0; n < NROW; ++n) {
                        acc[bb][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[n], acc[bb][m][n], 0, 0, 0);   // w_lo * x_hi
                        acc[bb][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[n], acc[bb][m][n], 0, 0, 0);   // w_hi * x_lo
                        acc[bb][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[n], acc[bb][m][n], 0, 0, 0);   // w_hi * x_hi
                    }
                }
            }
        }

        if (ph + 1 < nph) store_weights((ph + 1) & 1);
        if (u == 1 && k + 1 < P.NK) store_input((k + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: + bias (+ residual); lane owns output px (2X, 2X+1) of row 2Y + a: one float2 per cout
    const size_t HW = (size_t)P.H * P.W;
    const int xi = x0 + l31;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int cbase = cb * BM + (wm * 2 + m) * 32 + 4 * kg;
        float bq[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = cbase + (q & 3) + 8 * (q >> 2);
            bq[q] = P.bias ? P.bias[co < P.Cout ? co : P.Cout - 1] : 0.0f;
        }
#pragma unroll
        for (int n = 0; n < NROW; ++n) {
            const int yi = y0 + wr * NROW + n;
            if (yi < P.Hin && xi < P.Win) {
                const size_t o0 = ((size_t)b * P.Cout) * HW + (size_t)(2 * yi + a) * P.W + 2 * xi;
                float2 rq[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = cbase + (q & 3) + 8 * (q >> 2);
                    rq[q] = P.res ? *reinterpret_cast<const float2*>(P.res + o0 + (size_t)(co < P.Cout ? co : P.Cout - 1) * HW)
                                  : make_float2(0.0f, 0.0f);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = cbase + (q & 3) + 8 * (q >> 2);
                    if (co < P.Cout)
                        *reinterpret_cast<float2*>(P.y + o0 + (size_t)co * HW) =
                            make_float2(acc[0][m][n][q] + bq[q] + rq[q].x, acc[1][m][n][q] + bq[q] + rq[q].y);
                }
            }
        }
    }
}

// OIHW fp32 -> merged-tap records [a][cb][k][u][hl][b][v][mt][lane] of 8 bf16 (see k_upconv_bf16x3): the taps of a row
// parity a / tap u are dy in {0} | {1,2} (a = 0) or {0,1} | {2} (a = 1); columns alike.  The merged weight is the fp32 sum
// in (dy, dx) order.
__global__ void k_upconv_pack_bf16x3(const float* __restrict__ w, u32x4* __restrict__ out, int Cout, int Cin, int MT, int NCB, int NK, int perm) {
    const size_t n = (size_t)2 * NCB * NK * 2 * 2 * 2 * 2 * MT * 64;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t r = i;
    const int lane = (int)(r % 64); r /= 64;
    const int mt = (int)(r % MT); r /= MT;
    const int v = (int)(r % 2); r /= 2;
    const int bb = (int)(r % 2); r /= 2;
    const int hl = (int)(r % 2); r /= 2;
    const int u = (int)(r % 2); r /= 2;
    const int k = (int)(r % NK); r /= NK;
    const int cb = (int)(r % NCB); r /= NCB;
    const int a = (int)r;
    const int co = cb * MT * 32 + mt * 32 + (lane & 31);
    // tap sets: parity p, tap t -> [lo, hi) over the 3x3 index
    const int dy_lo = a == 0 ? (u == 0 ? 0 : 1) : (u == 0 ? 0 : 2), dy_hi = a == 0 ? (u == 0 ? 1 : 3) : (u == 0 ? 2 : 3);
    const int dx_lo = bb == 0 ? (v == 0 ? 0 : 1) : (v == 0 ? 0 : 2), dx_hi = bb == 0 ? (v == 0 ? 1 : 3) : (v == 0 ? 2 : 3);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = kstep_c0(k, lane >> 5, perm) + kstep_cj(j, perm);
        float val = 0.0f;
        if (co < Cout && ci < Cin)
            for (int dy = dy_lo; dy < dy_hi; ++dy)
                for (int dx = dx_lo; dx < dx_hi; ++dx) val += w[(((size_t)co * Cin + ci) * 3 + dy) * 3 + dx];
        const __bf16 h = (__bf16)val;
        o[j] = hl == 0 ? h : (__bf16)(val - (float)h);
    }
    out[i] = __builtin_bit_cast(u32x4, o);
}

inline int round_up_i(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

namespace mdt {

// shapes the split-bf16 3x3 kernel takes; everything else stays on the exact-fp32 kernel
bool conv_bf16x3_eligible(int cout, int cin, int ksize) { return ksize == 3 && cin % 16 == 0 && cout >= 32; }
// cout tiles per block: 128-cout blocks for the wide convs, 64-cout blocks for the small decoders' narrow ones
static int conv_bf16x3_mt(int cout) { return cout <= 64 ? 2 : 4; }
// K-step channel order (kstep_c0 / kstep_cj): the accumulator-lane order whenever whole 32-channel groups exist
static int conv_bf16x3_perm(int cin) { return cin % 32 == 0 ? 1 : 0; }

// record image = [ direct 3x3 records (9 taps) | sub-pixel upsample records (4 parities x 4 merged taps) ]
size_t conv_bf16x3_direct_records(int cout, int cin) {
    const int MT = conv_bf16x3_mt(cout), NCB = round_up_i(cout, MT * 32) / (MT * 32), NK = cin / 16;
    return (size_t)NCB * NK * 3 * 2 * 3 * MT * 64;
}
static size_t upconv_records(int cout, int cin) {
    const int MT = conv_bf16x3_mt(cout), NCB = round_up_i(cout, MT * 32) / (MT * 32), NK = cin / 16;
    return (size_t)2 * NCB * NK * 2 * 2 * 2 * 2 * MT * 64;
}
size_t conv_bf16x3_packed_floats(int cout, int cin) {   // size of the record array in floats (4 per 16-byte record)
    return (conv_bf16x3_direct_records(cout, cin) + upconv_records(cout, cin)) * 4;
}

int conv_bf16x3_pack(const float* d_w_oihw, void* d_out, int cout, int cin, hipStream_t s) {
    const int MT = conv_bf16x3_mt(cout), NCB = round_up_i(cout, MT * 32) / (MT * 32), NK = cin / 16, perm = conv_bf16x3_perm(cin);
    const size_t n = conv_bf16x3_direct_records(cout, cin);
    hipLaunchKernelGGL(k_conv_pack_bf16x3, dim3(cdiv((long long)n, 256)), dim3(256), 0, s, d_w_oihw, (u32x4*)d_out, cout, cin, MT, NCB, NK, perm);
    MDT_LAUNCH_CHECK();
    const size_t nu = upconv_records(cout, cin);
    hipLaunchKernelGGL(k_upconv_pack_bf16x3, dim3(cdiv((long long)nu, 256)), dim3(256), 0, s, d_w_oihw, (u32x4*)d_out + n, cout, cin, MT, NCB, NK, perm);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

// narrow convs (cout < 32: conv_out) only exist as a record-image kernel (vae_conv_rec.hip, one 32-cout tile per block): their
// packed image is the direct 3x3 records with MT = 1, NCB = 1, permuted K order
bool conv_rec_narrow_eligible(int cout, int cin, int ksize) { return ksize == 3 && cin % 32 == 0 && cout >= 1 && cout < 32; }
size_t conv_rec_narrow_packed_floats(int cin) { return (size_t)(cin / 16) * 3 * 2 * 3 * 1 * 64 * 4; }
int conv_rec_narrow_pack(const float* d_w_oihw, void* d_out, int cout, int cin, hipStream_t s) {
    const int NK = cin / 16;
    const size_t n = (size_t)NK * 3 * 2 * 3 * 64;
    hipLaunchKernelGGL(k_conv_pack_bf16x3, dim3(cdiv((long long)n, 256)), dim3(256), 0, s, d_w_oihw, (u32x4*)d_out, cout, cin, 1, 1, NK, 1);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

bool conv_bf16x3_gn_supported(int cout, int cin, int ksize, int up) {
    return conv_bf16x3_eligible(cout, cin, ksize) && !up && cin <= MAX_GN_CIN;
}

// statistics in the epilogue (k_conv3x3_bf16x3<4, true, 1, true>): 128-cout blocks with the fused pre-activation, whole blocks of couts
bool conv_bf16x3_stats_supported(int cout, int cin, int ksize, int up) {
    return conv_bf16x3_gn_supported(cout, cin, ksize, up) && conv_bf16x3_mt(cout) == 4 && cout % 128 == 0;
}
// doubles of the per-block partials: [B][ptiles][NCB][32 quads][2]
size_t conv_bf16x3_stats_part_doubles(int B, int cout, int H, int W) {
    return (size_t)B * ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * (cout / 128) * 64;
}

int conv_bf16x3_launch(const float* d_x, const void* d_w_rec, const float* d_bias, const float* d_res, float* d_y, int B, int cin,
                       int cout, int H, int W, int up, const float* d_coef, hipStream_t s, double* d_part) {
    ConvBParams P;
    P.perm = conv_bf16x3_perm(cin);
    P.coef = d_coef;
    P.gn_part = d_part;
    P.x = d_x; P.w = (const u32x4*)d_w_rec; P.bias = d_bias; P.res = d_res; P.y = d_y;
    P.B = B; P.Cin = cin; P.Cout = cout; P.H = H; P.W = W;
    P.Hin = up ? H / 2 : H; P.Win = up ? W / 2 : W; P.up = up;
    const int MT = conv_bf16x3_mt(cout);
    P.NCB = round_up_i(cout, MT * 32) / (MT * 32);
    P.NK = cin / 16;
    dim3 block(512);
    if (up) {   // fused nearest-2x: sub-pixel form (four 2x2 convs on the input grid)
        P.w = (const u32x4*)d_w_rec + conv_bf16x3_direct_records(cout, cin);
        P.PX = (P.Win + TW - 1) / TW;
        P.ptiles = P.PX * ((P.Hin + TH - 1) / TH);
        dim3 grid(((P.ptiles + 7) / 8) * 8 * P.NCB * 2, B);
        if (MT == 4) hipLaunchKernelGGL(k_upconv_bf16x3<4>, grid, block, 0, s, P);
        else hipLaunchKernelGGL(k_upconv_bf16x3<2>, grid, block, 0, s, P);
        MDT_LAUNCH_CHECK();
        return MDTILE_OK;
    }
    P.PX = (W + TW - 1) / TW;
    P.ptiles = P.PX * ((H + TH - 1) / TH);
    dim3 grid(((P.ptiles + 7) / 8) * 8 * P.NCB, B);
    MDT_CHECK_ARG(!d_part || (MT == 4 && d_coef && cout % 128 == 0), "conv_bf16x3_launch: no statistics kernel for cout=%d", cout);
    if (MT == 4) {
        if (d_coef && d_part) hipLaunchKernelGGL((k_conv3x3_bf16x3<4, true, 1, true>), grid, block, 0, s, P);
        else if (d_coef) hipLaunchKernelGGL((k_conv3x3_bf16x3<4, true>), grid, block, 0, s, P);
        else hipLaunchKernelGGL((k_conv3x3_bf16x3<4, false>), grid, block, 0, s, P);
    } else {
        if (d_coef) hipLaunchKernelGGL((k_conv3x3_bf16x3<2, true>), grid, block, 0, s, P);
        else hipLaunchKernelGGL((k_conv3x3_bf16x3<2, false>), grid, block, 0, s, P);
    }
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

// ldm Downsample (encoder): conv3x3 stride 2 over pad(x, right 1, bottom 1); output (Hin - 2) / 2 + 1 rows / columns
int conv_bf16x3_down2_launch(const float* d_x, const void* d_w_rec, const float* d_bias, float* d_y, int B, int cin, int cout, int Hin, int Win,
                             hipStream_t s) {
    ConvBParams P;
    P.perm = conv_bf16x3_perm(cin);
    P.coef = nullptr;
    P.gn_part = nullptr;
    P.x = d_x; P.w = (const u32x4*)d_w_rec; P.bias = d_bias; P.res = nullptr; P.y = d_y;
    P.B = B; P.Cin = cin; P.Cout = cout; P.H = (Hin - 2) / 2 + 1; P.W = (Win - 2) / 2 + 1;
    P.Hin = Hin; P.Win = Win; P.up = 0;
    const int MT = conv_bf16x3_mt(cout);
    P.NCB = round_up_i(cout, MT * 32) / (MT * 32);
    P.NK = cin / 16;
    P.PX = (P.W + TW - 1) / TW;
    P.ptiles = P.PX * ((P.H + TH - 1) / TH);
    dim3 grid(((P.ptiles + 7) / 8) * 8 * P.NCB, B), block(512);
    if (MT == 4) hipLaunchKernelGGL((k_conv3x3_bf16x3<4, false, 2>), grid, block, 0, s, P);
    else hipLaunchKernelGGL((k_conv3x3_bf16x3<2, false, 2>), grid, block, 0, s, P);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

}  // namespace mdt
