// 1x1 conv tiles of the Tiled-VAE task queue (nin_shortcut of the channel-changing ResnetBlocks, q / k / proj_out of the
// mid-block attention; scripts/tilevae.py:115-137, tile_utils/attn.py:50-70) on the bf16 matrix cores with SPLIT-fp32
// operands -- the same arithmetic contract as vae_conv_bf16x3.hip (x = hi + lo, three bf16 MFMAs per product, fp32
// accumulation, ~1e-5 relative to fp32).  The exact-fp32 kernel (vae_conv.hip) remains behind MDTILE_CONV_EXACT_F32.
//
// A 1x1 conv is point-wise in space, so the image is treated as ONE flat run of H*W pixels: a block owns 256 consecutive
// pixels (8 MFMA column tiles of 32) x BM couts; every global load is a full 256-byte row per wave, there are no halos and
// no ragged rows.  GEMM view: D[cout][px] = sum_cin W[cout][cin] X[cin][px], MFMA v_mfma_f32_32x32x16_bf16 with
// A = weights (M = 32 couts, K = 16 cin), B = input (K = 16 cin, N = 32 px).
//   phase = 32 input channels (two 16-channel K-steps), ONE barrier per phase, register-prefetched double-buffered LDS:
//     input  LDS image  [hl][ks 2][kg 2][px 256] x 16 B   (record = 8 channels of one pixel)
//     weight LDS image  [hl][ks 2][mtile][lane 64] x 16 B  (exactly the pre-packed global order: straight copy)
//   every fragment read is a conflict-free ds_read_b128 of 32 consecutive records per half-wave.
// Block = 512 threads = 8 waves; wave = 64 couts x NCOL column tiles.
// These convs are HBM-bound on the wide images (nin_shortcut 256 -> 128 at 2224^2: 7.6 GB against 0.32 TFLOP), so (round 3)
//   * the operands of phase ph + 2 are requested while phase ph computes (two register sets): two phases = 64 KB of input per CU
//     in flight instead of one (measured before: 3.4 TB/s of traffic at one phase in flight, ~6 B/clk/CU);
//   * BM goes up to 256 couts (MT = 8: eight accumulator tiles per wave): 512 -> 256 reads its input once instead of twice.
#include <type_traits>

#include "common.h"

using namespace mdt;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct Conv1Params {
    const float* x;      // [B, Cin, HW] fp32
    const u32x4* w;      // packed bf16 hi/lo records, see k_conv1x1_pack_bf16x3
    const float* bias;   // [Cout] or null
    const float* res;    // residual [B, Cout, HW] or null
    float* y;            // [B, Cout, HW]
    int B, Cin, Cout;
    size_t HW;
    int ptiles, NCB, NP; // 256-pixel tiles, cout blocks, 32-channel phases
};


__device__ __forceinline__ void split8c(const float (&v)[8], u32x4& hi, u32x4& lo) {
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = (__bf16)v[i];
        l[i] = (__bf16)(v[i] - (float)h[i]);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

// PXT = pixels per block.  128-cout blocks take 128 px: 64 KB of LDS and ~100 registers -> TWO blocks per CU, so that the store-bound
// epilogue of one block (128 couts x 128 px x 4 B through 4-byte-per-lane stores) runs under the K loop of the other; with one
// resident block per CU the epilogue was fully exposed (256 -> 128 at 2224^2: ~30 us per block of which ~half epilogue).
template <int MT, int PXT>
__global__ __launch_bounds__(512, PXT == 128 ? 4 : 2) void k_conv1x1_bf16x3(const Conv1Params P) {
    constexpr int BM = MT * 32;
    constexpr int IN_REC1 = 2 * 2 * PXT;             // records per hl per stage: [ks][kg][px]
    constexpr int NG = PXT / 128;                    // 8-channel groups a thread stages per phase (512 threads x NG = 4 groups x PXT px)
    constexpr int WAVES_M = MT / 2, WAVES_C = 8 / WAVES_M, NCOL = (PXT / 32) / WAVES_C;   // column tiles (32 px) per wave
    static_assert(NCOL >= 1 && NG >= 1, "block shape");
    constexpr int W_REC = 2 * 2 * MT * 64;           // [hl][ks][mt][lane]
    constexpr int NWREG = W_REC / 512;               // 4 (MT = 8), 2 (MT = 4) or 1 (MT = 2)
    constexpr int IN_STAGE = 2 * IN_REC1;
    __shared__ u32x4 smem[2 * IN_STAGE + 2 * W_REC];
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + 2 * IN_STAGE;

    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int ptile = (slot / P.NCB) * 8 + xcd, cb = slot % P.NCB;
    if (ptile >= P.ptiles) return;
    const int b = blockIdx.y;
    const size_t p0 = (size_t)ptile * PXT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kg = lane >> 5;
    const int wm = wave % WAVES_M, wc = wave / WAVES_M;
    const float* xb = P.x + (size_t)b * P.Cin * P.HW;

    // staging map: thread -> pixel (tid % PXT) and NG consecutive 8-channel groups gi = (tid / PXT) * NG + g of the phase's four
    // (gi = 2 * K-step + kg: channels 32 ph + 8 gi + j, LDS record gi * PXT + px)
    const int spx = tid & (PXT - 1), sgb = (tid / PXT) * NG;
    const bool pin = p0 + spx < P.HW;
    const size_t soff = pin ? p0 + spx : 0;
    float rin[2][NG][8];         // [register set][8-channel group][channel]
    u32x4 rwt[2][NWREG];

    auto load_input = [&](int set, int ph) {       // phase ph: channels 32 ph .. 32 ph + 31
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float* src = xb + (size_t)(ph * 32 + (sgb + g) * 8) * P.HW + soff;
#pragma unroll
            for (int j = 0; j < 8; ++j) rin[set][g][j] = src[(size_t)j * P.HW];
        }
    };
    auto store_input = [&](int set, int stage) {
        u32x4* dst = in_l + stage * IN_STAGE;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pin ? rin[set][g][j] : 0.0f;
            u32x4 hi, lo;
            split8c(v, hi, lo);
            const int rec = (sgb + g) * PXT + spx;
            dst[rec] = hi;
            dst[IN_REC1 + rec] = lo;
        }
    };
    const u32x4* wsrc = P.w + (size_t)cb * P.NP * W_REC;
    auto load_weights = [&](int set, int ph) {
        const u32x4* src = wsrc + (size_t)ph * W_REC;
#pragma unroll
        for (int i = 0; i < NWREG; ++i) rwt[set][i] = src[tid + 512 * i];
    };
    auto store_weights = [&](int set, int stage) {
        u32x4* dst = w_l + stage * W_REC;
#pragma unroll
        for (int i = 0; i < NWREG; ++i) dst[tid + 512 * i] = rwt[set][i];
    };

    f32x16 acc[2][NCOL];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NCOL; ++n)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.0f;

    load_input(0, 0);
    load_weights(0, 0);
    if (P.NP > 1) {
        load_input(1, 1);
        load_weights(1, 1);
    }
    store_input(0, 0);
    store_weights(0, 0);
    __syncthreads();

    // phase ph sits in LDS stage ph & 1 and came through register set ph & 1; while it computes, phase ph + 2 is requested into the
    // same register set (free since its contents went to LDS) and phase ph + 1 -- requested a whole phase ago -- is written to LDS
    // behind the MFMAs.  Two phases per trip keep the register-set index a compile-time constant.
    auto phase = [&](int ph, auto set_tag) {
        constexpr int set = decltype(set_tag)::value;
        if (ph + 2 < P.NP) {
            load_input(set, ph + 2);
            load_weights(set, ph + 2);
        }
        const u32x4* wst = w_l + set * W_REC;
        const u32x4* ist = in_l + set * IN_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[2][2];   // [m][hl]
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    a[m][hl] = __builtin_bit_cast(bf16x8, wst[((hl * 2 + ks) * MT + wm * 2 + m) * 64 + lane]);
#pragma unroll
            for (int n = 0; n < NCOL; ++n) {
                const int rec = (ks * 2 + kg) * PXT + (wc * NCOL + n) * 32 + l31;
                const bf16x8 bh = __builtin_bit_cast(bf16x8, ist[rec]), bl = __builtin_bit_cast(bf16x8, ist[IN_REC1 + rec]);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], bh, acc[m][n], 0, 0, 0);   // w_lo * x_hi
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], bl, acc[m][n], 0, 0, 0);   // w_hi * x_lo
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], bh, acc[m][n], 0, 0, 0);   // w_hi * x_hi
                }
            }
        }
        if (ph + 1 < P.NP) {
            store_weights(set ^ 1, set ^ 1);
            store_input(set ^ 1, set ^ 1);
        }
        __syncthreads();
    };
    for (int ph = 0; ph < P.NP; ph += 2) {
        phase(ph, std::integral_constant<int, 0>{});
        if (ph + 1 < P.NP) phase(ph + 1, std::integral_constant<int, 1>{});
    }

    // epilogue: + bias (+ residual).  C/D layout of a 32x32 MFMA: col = lane & 31, row = (q&3) + 8*(q>>2) + 4*(lane>>5)
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
        for (int n = 0; n < NCOL; ++n) {
            const size_t p = p0 + (wc * NCOL + n) * 32 + l31;
            if (p < P.HW) {
                const size_t o0 = ((size_t)b * P.Cout) * P.HW + p;
                float rq[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = cbase + (q & 3) + 8 * (q >> 2);
                    rq[q] = P.res ? P.res[o0 + (size_t)(co < P.Cout ? co : P.Cout - 1) * P.HW] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = cbase + (q & 3) + 8 * (q >> 2);
                    if (co < P.Cout) P.y[o0 + (size_t)co * P.HW] = acc[m][n][q] + bq[q] + rq[q];
                }
            }
        }
    }
}

// OI fp32 -> records [cb][phase][hl][ks][mt][lane] of 8 bf16: cout = cb*BM + mt*32 + (lane & 31),
// cin = phase*32 + ks*16 + (lane >> 5)*8 + j.  Zero outside [Cout).
__global__ void k_conv1x1_pack_bf16x3(const float* __restrict__ w, u32x4* __restrict__ out, int Cout, int Cin, int MT, int NCB, int NP) {
    const size_t n = (size_t)NCB * NP * 2 * 2 * MT * 64;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t r = i;
    const int lane = (int)(r % 64); r /= 64;
    const int mt = (int)(r % MT); r /= MT;
    const int ks = (int)(r % 2); r /= 2;
    const int hl = (int)(r % 2); r /= 2;
    const int ph = (int)(r % NP); r /= NP;
    const int cb = (int)r;
    const int co = cb * MT * 32 + mt * 32 + (lane & 31);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = ph * 32 + ks * 16 + (lane >> 5) * 8 + j;
        const float v = (co < Cout && ci < Cin) ? w[(size_t)co * Cin + ci] : 0.0f;
        const __bf16 h = (__bf16)v;
        o[j] = hl == 0 ? h : (__bf16)(v - (float)h);
    }
    out[i] = __builtin_bit_cast(u32x4, o);
}

inline int round_up1(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

namespace mdt {

bool conv1x1_bf16x3_eligible(int cout, int cin) { return cin % 32 == 0 && cout >= 32; }
static int conv1x1_mt(int cout) {
    // MDTILE_C1X1_MT8=0 (probing, read once: packing and launch must agree): 128-cout blocks also for wider convs
    static const bool mt8 = [] { const char* e = getenv("MDTILE_C1X1_MT8"); return !(e && e[0] == '0'); }();
    return cout > 128 && mt8 ? 8 : (cout > 64 ? 4 : 2);
}

size_t conv1x1_bf16x3_packed_floats(int cout, int cin) {
    const int MT = conv1x1_mt(cout), NCB = round_up1(cout, MT * 32) / (MT * 32), NP = cin / 32;
    return (size_t)NCB * NP * 2 * 2 * MT * 64 * 4;
}

int conv1x1_bf16x3_pack(const float* d_w_oihw, void* d_out, int cout, int cin, hipStream_t s) {
    const int MT = conv1x1_mt(cout), NCB = round_up1(cout, MT * 32) / (MT * 32), NP = cin / 32;
    const size_t n = (size_t)NCB * NP * 2 * 2 * MT * 64;
    hipLaunchKernelGGL(k_conv1x1_pack_bf16x3, dim3(cdiv((long long)n, 256)), dim3(256), 0, s, d_w_oihw, (u32x4*)d_out, cout, cin, MT, NCB, NP);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

int conv1x1_bf16x3_launch(const float* d_x, const void* d_w_rec, const float* d_bias, const float* d_res, float* d_y, int B, int cin,
                          int cout, size_t HW, hipStream_t s) {
    Conv1Params P;
    P.x = d_x; P.w = (const u32x4*)d_w_rec; P.bias = d_bias; P.res = d_res; P.y = d_y;
    P.B = B; P.Cin = cin; P.Cout = cout; P.HW = HW;
    const int MT = conv1x1_mt(cout);
    const int PXT = MT == 4 ? 128 : 256;
    P.ptiles = (int)((HW + PXT - 1) / PXT);
    P.NCB = round_up1(cout, MT * 32) / (MT * 32);
    P.NP = cin / 32;
    dim3 grid(((P.ptiles + 7) / 8) * 8 * P.NCB, B), block(512);
    if (MT == 8) hipLaunchKernelGGL((k_conv1x1_bf16x3<8, 256>), grid, block, 0, s, P);
    else if (MT == 4) hipLaunchKernelGGL((k_conv1x1_bf16x3<4, 128>), grid, block, 0, s, P);
    else hipLaunchKernelGGL((k_conv1x1_bf16x3<2, 256>), grid, block, 0, s, P);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

}  // namespace mdt
