// Shared pieces of the record-image conv kernels (vae_conv_rec.hip: one 8-wave block per CU; vae_conv_rec2.hip: two independent
// 4-wave blocks per CU): kernel parameters, the inline-asm LDS-DMA, the split, the shared epilogue and the LDS input stage.
// See vae_conv_rec.hip for the record-image format and the upstream call sites (scripts/tilevae.py:115-195, 218-245, 614-616).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace mdt {

// Geometry of a record image (round 4).  A plane has H + 2 rows (one zero-border row above and below) of rec_pitch(W) records;
// pixel x sits at column x + 8, the zero-border records at columns 7 and W + 8; columns 0-6 and those behind W + 8 are padding that
// nothing reads.  The pitch is a multiple of 8 records, so every 32-pixel run a wave stores (x0 % 32 == 0) is FOUR WHOLE 128-byte lines.
// Until round 4 the interior started at column 1 of a W + 2 pitch: every run began 16 bytes into a line and ended 16 bytes into
// another, i.e. two partial 64-byte sectors per 512 bytes written (probes/cu_mempipe_probe.cpp: 2 x 512 B runs shifted by 16 B stream
// at 15 GB/s per CU chip-wide, aligned ones at 28).  Measured on the conv itself the aligned rows are worth 1-2 % of an item
// (profiles/r4n): its store epilogue is bound elsewhere, but whole-line writes are the right layout anyway.
constexpr int REC_COL0 = 7;                                                            // column of the left border record
__host__ __device__ inline int rec_pitch(int W) { return (W + 2 + REC_COL0 + 7) & ~7; }

constexpr int REC_WIN_MAXB = 8;   // images per launch that may carry a window origin of their own (stacked tiles of one shape)

struct ConvRParams {
    const u32x4* x;      // input record image [B][2][Cin/8][Hin+2][Win+2]
    const u32x4* w;      // packed weights (vae_conv_bf16x3.hip: k_conv_pack_bf16x3 / k_upconv_pack_bf16x3, permuted K order)
    const float* bias;   // [Cout] or null
    const float* res;    // residual [B, Cout, H, W] fp32 or null
    float* y32;          // fp32 output [B, Cout, H, W] or null
    u32x4* yrec;         // record-image output [B][2][Cout/8][H+2][W+2] or null
    const float* coef;   // activation of the record output: [B][2][Cout] = (a, s), yrec = split(silu(a y + s)); null = split(y)
    int B, Cin, Cout, H, W;   // H, W: OUTPUT size
    int Hin, Win;             // input size (= H, W; half of it for the sub-pixel upsample kernel)
    int HinF, WinF;           // sub-pixel upsample kernel: image b's input is the window [iy0[b] : iy0[b] + Hin, ix0[b] : ix0[b] + Win] of a
    int iy0[REC_WIN_MAXB], ix0[REC_WIN_MAXB];   // record image of HinF x WinF px (whole image: HinF = Hin, WinF = Win, all origins 0)
    int ptiles, PX, NCB, NK;  // pixel tiles, tiles per row, cout blocks, 16-channel K-steps
    // two-blocks-per-CU kernels (vae_conv_rec2.hip) only:
    unsigned skew_ticks;      // start-up delay (100 MHz ticks) of the block that arrives SECOND on its CU: puts the pair half an item out of phase
    unsigned* cu_ctr;         // [8 XCDs x 256 hardware CU ids] arrival counters (never reset: only the parity is used); null = skew by block index
    unsigned* census;         // probing: [gridDim.x] hardware id of the CU each block ran on | arrival parity << 31, or null
    int dbg;                  // probing (MDTILE_REC_DBG): bit 0 = skip the epilogue (K loop only: nothing is written); bit 3 = block 0 of the
                              // one-block kernel writes s_memtime stamps per wave and item to `census` (probes/conv_item_timeline.py)
};

}  // namespace mdt

namespace {

using mdt::ConvRParams;
using mdt::REC_WIN_MAXB;

__device__ __forceinline__ void split8r(const float (&v)[8], u32x4& hi, u32x4& lo) {
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = (__bf16)v[i];
        l[i] = (__bf16)(v[i] - (float)h[i]);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

// LDS-DMA of one 16-byte record per lane: global (scalar base + 32-bit lane offset) -> LDS (wave-uniform base + 16 * lane).
// Issued through inline asm on purpose: hipcc books a __builtin_amdgcn_global_load_lds as a pending FLAT access and then
// degrades EVERY later `s_waitcnt lgkmcnt(N)` to lgkmcnt(0) -- the fragment prefetch below would wait for the reads it has
// just issued.  The asm is invisible to that bookkeeping; its completion is counted by hand (vmcnt(0) + barrier before any
// ds_read of the data).  M0 = LDS destination, restored afterwards (compiler-reserved); s_nop: M0 / SGPR-base write -> VMEM read.
__device__ __forceinline__ void dma16(const char* base, unsigned voff, const u32x4* lds_dst) {
    const unsigned l = (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) u32x4*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(l)
                 : "memory");
}

__device__ __forceinline__ float silu_f(float t) { return t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)); }

// ---- shared epilogue: one 32-cout tile of a wave (NROW pixel rows x NPX pixels per lane) -> + bias (+ residual) -> fp32 NCHW
// and / or record image.  C/D layout of a 32x32 MFMA: col = lane & 31 (pixel), row = (q&3) + 8*(q>>2) + 4*(lane>>5) (cout).
// The epilogue is latency-, not bandwidth-bound (one block per CU, nothing else to run meanwhile), so it is built to expose
// as few memory round trips as possible:
//   * the per-channel constants (bias, and the (a, s) of the record output's activation) of the item's 128 couts are DMA'd
//     into a 3 x 1 KB LDS buffer together with the item's first operands -- the epilogue reads them with ds_read_b128;
//   * the residual is requested 32 values per lane at a time (two pixel rows) before the first of them is used.
struct EpiCtx {
    const float* __restrict__ res;
    float* __restrict__ y32;
    u32x4* __restrict__ yrec;
    bool has_bias, has_act;
    int Cout, H, W;      // output size
    int b, kg;
    size_t HW, planeO;   // fp32 plane, record plane ((H + 2) * rec_pitch(W))
    int WpO;
};

constexpr int EC_REC = 3 * 64;   // records of one constants buffer: [bias | a | s] x 1 KB (128 floats + pad for the DMA's upper lanes)

// NPX = 1: one pixel per lane; NPX = 2: the lane owns output px (2X, 2X+1) (sub-pixel upsample kernel)
// ECS: float4 stride between the [bias | a | s] slots of the constants buffer (64 = 1 KB slots; 32 = packed 512 B slots)
template <int NPX, int NROW, int ECS = 64>
__device__ __forceinline__ void epilogue_mtile(const EpiCtx& E, const u32x4* ec, f32x16 (&acc)[NROW][NPX], int mt_local, int mt_global,
                                               const int (&ys)[NROW], int x, bool x_ok) {
    // constants of this lane's 16 couts: q = 4 g + i  <->  channel 32 mt + 4 kg + 8 g + i
    float bq[16], aq[16], sq[16];
    {
        const float4* e4 = reinterpret_cast<const float4*>(ec);
        const int c4 = (mt_local * 32 + 4 * E.kg) >> 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 tb = E.has_bias ? e4[c4 + 2 * g] : make_float4(0.f, 0.f, 0.f, 0.f);
            bq[4 * g] = tb.x; bq[4 * g + 1] = tb.y; bq[4 * g + 2] = tb.z; bq[4 * g + 3] = tb.w;
            if (E.has_act) {
                const float4 ta = e4[ECS + c4 + 2 * g], ts = e4[2 * ECS + c4 + 2 * g];
                aq[4 * g] = ta.x; aq[4 * g + 1] = ta.y; aq[4 * g + 2] = ta.z; aq[4 * g + 3] = ta.w;
                sq[4 * g] = ts.x; sq[4 * g + 1] = ts.y; sq[4 * g + 2] = ts.z; sq[4 * g + 3] = ts.w;
            }
        }
    }
    const int cbase = mt_global * 32 + 4 * E.kg;
    const size_t obase = ((size_t)E.b * E.Cout + cbase) * E.HW;
    const int xc = x_ok ? x : 0;                                  // clamped column for the unconditional residual loads
#pragma unroll
    for (int n = 0; n < NROW; ++n)
#pragma unroll
        for (int e = 0; e < NPX; ++e)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[n][e][q] += bq[q];
    if (E.res) {
        constexpr int RB = NPX == 1 ? 2 : 1;      // rows whose residual is in flight together (32 registers)
#pragma unroll
        for (int n0 = 0; n0 < NROW; n0 += RB) {
            float r[RB][NPX][16];
#pragma unroll
            for (int n = 0; n < RB; ++n) {
                const int yc = ys[n0 + n] < E.H ? ys[n0 + n] : E.H - 1;
                const float* rp0 = E.res + obase + (size_t)yc * E.W + xc;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float* rp = rp0 + (size_t)((q & 3) + 8 * (q >> 2)) * E.HW;
                    if (NPX == 2) {
                        const float2 r2 = *reinterpret_cast<const float2*>(rp);
                        r[n][0][q] = r2.x;
                        r[n][NPX - 1][q] = r2.y;
                    } else {
                        r[n][0][q] = *rp;
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < RB; ++n)
#pragma unroll
                for (int e = 0; e < NPX; ++e)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[n0 + n][e][q] += r[n][e][q];
        }
    }
#pragma unroll
    for (int n = 0; n < NROW; ++n) {
        const int y = ys[n];
        if (!(y < E.H && x_ok)) continue;
        const size_t o0 = obase + (size_t)y * E.W + x;
        if (E.y32) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (cbase + (q & 3) + 8 * (q >> 2) >= E.Cout) continue;          // narrow convs: couts past Cout are padding
                float* yp = E.y32 + o0 + (size_t)((q & 3) + 8 * (q >> 2)) * E.HW;
                if (NPX == 2) *reinterpret_cast<float2*>(yp) = make_float2(acc[n][0][q], acc[n][NPX - 1][q]);
                else *yp = acc[n][0][q];
            }
        }
        if (E.yrec) {
            // records R = 0 (q 0..7) and R = 1 (q 8..15) of this lane: planes ((mt*2 + R)*2 + kg)
            const int Pn = E.Cout >> 3;
            u32x4* yb = E.yrec + (size_t)E.b * 2 * Pn * E.planeO;
#pragma unroll
            for (int R = 0; R < 2; ++R) {
                const size_t pl = (size_t)((mt_global * 2 + R) * 2 + E.kg) * E.planeO;
#pragma unroll
                for (int e = 0; e < NPX; ++e) {
                    float t8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float t = acc[n][e][8 * R + j];
                        t8[j] = E.has_act ? silu_f(fmaf(t, aq[8 * R + j], sq[8 * R + j])) : t;
                    }
                    u32x4 hi, lo;
                    split8r(t8, hi, lo);
                    const size_t at = pl + (size_t)(y + 1) * E.WpO + (x + e + 1 + mdt::REC_COL0);
                    yb[at] = hi;
                    yb[(size_t)Pn * E.planeO + at] = lo;
                }
                // zero border of the record image (this block owns the border cells next to its edge pixels)
                const bool left = x == 0, right = x + NPX == E.W, top = y == 0, bot = y == E.H - 1;
                if (left || right || top || bot) {
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    auto zrec = [&](int py, int px) {
                        const size_t at = pl + (size_t)py * E.WpO + px + mdt::REC_COL0;      // (px: padded column, 0 = left border)
                        yb[at] = z;
                        yb[(size_t)Pn * E.planeO + at] = z;
                    };
                    if (left) zrec(y + 1, 0);
                    if (right) zrec(y + 1, E.W + 1);
                    if (top) {
#pragma unroll
                        for (int e = 0; e < NPX; ++e) zrec(0, x + e + 1);
                        if (left) zrec(0, 0);
                        if (right) zrec(0, E.W + 1);
                    }
                    if (bot) {
#pragma unroll
                        for (int e = 0; e < NPX; ++e) zrec(E.H + 1, x + e + 1);
                        if (left) zrec(E.H + 1, 0);
                        if (right) zrec(E.H + 1, E.W + 1);
                    }
                }
            }
        }
    }
}

#define MDT_PIN() __builtin_amdgcn_sched_barrier(0)

// =====================================================================================================================
// LDS input stage shared by both kernels: [hl][kg][ROWS][34] records, each hl half padded to whole 64-record DMA pieces so
// that one wave-instruction never straddles the two halves (hl then sits in the scalar base address, the lane offset
// stays 32-bit: global_load_lds with saddr + voffset).
template <int ROWS, int NWAVES = 8>
struct InStage {
    static constexpr int COLS = 34;
    static constexpr int HALF = 2 * ROWS * COLS;               // records of one hl half
    static constexpr int HALF_DMA = (HALF + 63) / 64;           // wave-instructions per half
    static constexpr int HALF_PAD = HALF_DMA * 64;
    static constexpr int DMA = 2 * HALF_DMA, PAD = 2 * HALF_PAD;
    static constexpr int PW = (DMA + NWAVES - 1) / NWAVES;      // wave-instructions per wave
};

}  // namespace
