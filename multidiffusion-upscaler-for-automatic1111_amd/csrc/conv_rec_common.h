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
    unsigned* cu_ctr;         // [16 XCC ids x 256 hardware CU ids] arrival counters, word = launch epoch << 8 | blocks of that launch seen on the CU; null = skew by block index
    unsigned epoch;           // this launch's epoch (24 bits, host counter per device): a counter word of another epoch restarts at 0
    unsigned* census;         // probing: [gridDim.x] hardware id of the CU each block ran on | arrival parity << 31, or null
    double* gn_part;          // statistics kernels (k_conv3x3_rec_st / k_upconv_rec_st) only: per-wave partials of the output, see epilogue_item<.., ST>
    int dbg;                  // probing (MDTILE_REC_DBG): bit 0 = skip the epilogue (K loop only: nothing is written); bit 3 = block 0 of the
                              // one-block kernel writes s_memtime stamps per wave and item to `census` (probes/conv_item_timeline.py)
};

}  // namespace mdt

namespace {

using mdt::ConvRParams;
using mdt::REC_WIN_MAXB;

// ConvRParams::dbg as the kernels see it: the constant 0 in the shipping library (common.h: kProbes) -- every probe branch folds away
__device__ __forceinline__ int pdbg(int d) { return mdt::kProbes ? d : 0; }

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

// the same with the LDS destination given as a byte address (per-wave constant + compile-time offset: one s_add, no pointer arithmetic)
__device__ __forceinline__ void dma16(const char* base, unsigned voff, const u32x4*, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_byte_addr)
                 : "memory");
}

__device__ __forceinline__ float silu_f(float t) { return t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)); }

// ---- shared epilogue: one 32-cout tile of a wave (NROW pixel rows x NPX pixels per lane) -> + bias (+ residual) -> fp32 NCHW
// and / or record image.  C/D layout of a 32x32 MFMA: col = lane & 31 (pixel), row = (q&3) + 8*(q>>2) + 4*(lane>>5) (cout).
// The per-channel constants (bias, and the (a, s) of the record output's activation) of the item's 128 couts are DMA'd into a
// 3 x 1 KB LDS buffer together with the item's first operands -- the epilogue reads them with ds_read_b128 where it uses them.
// A conv2's residual does not pass through here at all in the one-pixel-per-lane kernels: it is in the accumulators (ResRows below).
struct EpiCtx {
    const float* __restrict__ res;
    float* __restrict__ y32;
    u32x4* __restrict__ yrec;
    bool has_bias, has_act;
    int Cout, H, W;      // output size
    int b, kg;
    size_t HW, planeO;   // fp32 plane, record plane ((H + 2) * rec_pitch(W))
    int WpO;
    double* st;          // ST epilogues: this wave's 16 (sum, sum of squares) slots of the item -- 8 quads of each of its two 32-cout tiles (wave-uniform)
    int dbg;             // probing (ConvRParams::dbg): bit 4 = the fp32 stores are skipped, bit 5 = the record stores are skipped (probes/conv_item_timeline.py --dbg)
};

constexpr int EC_REC = 3 * 64;   // records of one constants buffer: [bias | a | s] x 1 KB (128 floats + pad for the DMA's upper lanes)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// v <- silu(a v + s) for 8 of the 16 accumulator values of one pixel (one record's worth: o = 0 or 8): the same arithmetic as silu_f(fmaf(v, a, s)), value by value, but
// written stage by stage over its 8 (independent) chains and on packed fp32 pairs.  In the epilogue nothing else runs on the SIMD,
// so what the wave does not overlap itself is lost: with the chains interleaved two at a time (what hipcc made of the per-value form)
// every v_exp_f32 / v_rcp_f32 waited out its own latency.
__device__ __forceinline__ void act8(f32x16& v, int o, const f32x2 (&a)[4], const f32x2 (&s)[4]) {
    f32x2 t[4], e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = __builtin_elementwise_fma(f32x2{v[o + 2 * j], v[o + 2 * j + 1]}, a[j], s[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = t[j] * -1.44269504088896340736f;      // __expf(-t) = exp2(t * -log2(e))
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = f32x2{__builtin_amdgcn_exp2f(e[j].x), __builtin_amdgcn_exp2f(e[j].y)};
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = e[j] + 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = f32x2{__builtin_amdgcn_rcpf(e[j].x), __builtin_amdgcn_rcpf(e[j].y)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[j] = t[j] * e[j];
        v[o + 2 * j] = t[j].x;
        v[o + 2 * j + 1] = t[j].y;
    }
}

// 8 values -> (hi, lo) records, two values per instruction: v_cvt_pk_bf16_f32 rounds a PAIR; the pair's hi parts come back as fp32
// with one shift and one mask; lo = bf16(v - hi) as a packed subtract and a second packed convert (2.5 VALU per value; split8r's
// per-value form compiled to 4).
__device__ __forceinline__ void split8p(const f32x16& v, int o, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 p = {v[o + 2 * j], v[o + 2 * j + 1]};
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2));
        const f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
        hi[j] = h;
        lo[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(p - hf, bf16x2));
    }
}

// A wave-uniform pointer the optimiser cannot see through: what is added to it afterwards stays a 32-bit lane offset next to an SGPR
// base (LLVM otherwise re-associates base + plane + lane into (base + lane) + plane: a 64-bit VALU multiply-add per access).
// volatile: not merged across pixel rows either (16 live SGPR pairs per tensor would spill to VGPR lanes).
// Returned as a GLOBAL-address-space pointer (the asm hides where the value came from; a generic pointer would compile to flat_*).
#define MDT_GLOBAL __attribute__((address_space(1)))
typedef MDT_GLOBAL char gchar;
__device__ __forceinline__ gchar* uniform_ptr(const void* p) {
    size_t v = reinterpret_cast<size_t>(p);
    asm volatile("" : "+s"(v));
    return (gchar*)v;
}
__device__ __forceinline__ gchar* uniform_ptr(gchar* p) { return uniform_ptr((const void*)p); }
// the plane of accumulator value q is (q & 3) + 8 (q >> 2): walking q = 0..15 the uniform base advances by 1, 1, 1, 5, 1, 1, 1, 5, ... planes
#define MDT_PLANE_STEP(q) (((q) & 3) ? 1 : 5)

// ---- the residual of a conv2 arrives IN THE ACCUMULATORS (one-pixel-per-lane kernels).  The accumulator registers of an item are free from the
// moment their rows are stored; the residual rows of the block's NEXT item are loaded into them right there, and that item's K loop starts
// from them instead of from zero: y = ((res + sum of products) + bias).  The loads queue behind the stores in the CU's in-order memory pipe,
// but nobody waits for them before the item-top vmcnt(0) that waits for the stores anyway -- in the form that read the residual inside the
// epilogue every two-row unit stalled on a round trip stuck behind the previous unit's stores (36.5 k cycles against 21.1 k for the same
// item without a residual, profiles/r4x).  No buffer registers at all.  The first item of a block loads its rows before the loop.
template <int NROW>
struct ResRows {
    bool on;             // (wave-uniform) there is such an item
    int b, mt_global0;   // image, first 32-cout tile of the wave
    int ys[NROW];        // pixel rows of the wave
    int x;               // the lane's pixel column
    bool x_ok;
};

// rows n0 .. n0 + NR - 1 of tile m (clamped coordinates: unconditional loads, as in the epilogue form)
template <int NROW, int MW, int NR>
__device__ __forceinline__ void residual_into_acc(const float* res, int Cout, size_t HW, int H, int W, int kg, const ResRows<NROW>& R, int m, int n0,
                                                  f32x16 (&acc)[MW][NROW][1]) {
    const size_t HW4 = HW * sizeof(float);
    unsigned kgo = (unsigned)kg;
    asm volatile("" : "+v"(kgo));
    const unsigned lane32 = ((4u * kgo) * (unsigned)HW + (unsigned)(R.x_ok ? R.x : 0)) * 4u;
    const char* rb = reinterpret_cast<const char*>(res) + ((size_t)R.b * Cout + (size_t)(R.mt_global0 + m) * 32) * HW4;
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int yc = R.ys[n0 + n] < H ? R.ys[n0 + n] : H - 1;
        const unsigned ro = lane32 + (unsigned)(yc * W) * 4u;
        gchar* up = uniform_ptr(rb);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q) up = uniform_ptr(up + MDT_PLANE_STEP(q) * HW4);
            acc[m][n0 + n][0][q] = *(const MDT_GLOBAL float*)(up + (size_t)ro);
        }
    }
}

// NPX = 1: one pixel per lane; NPX = 2: the lane owns output px (2X, 2X+1) (sub-pixel upsample kernel)
// ECS: float4 stride between the [bias | a | s] slots of the constants buffer (64 = 1 KB slots; 32 = packed 512 B slots)
//
// Round 4, late (DESIGN.md section 3 "Late round 4"): hipcc's code for the first form of this epilogue spent ~100 VALU instructions per
// 8-value record (per-value converts, 64-bit address arithmetic per access, an activation computed and then selected away when there
// was none, a Cout test and five SGPR-spill reloads in front of every fp32 store) -- with two waves per SIMD that was as long as the
// store drain it should have hidden behind.  This form addresses every access as (wave-uniform 64-bit base) + (32-bit lane offset) --
// global_* with an SGPR base -- so an access costs scalar adds only; the uniform bases are formed per item behind an opaque asm, or
// LLVM hoists 16 of them per tensor out of the persistent loop and spills them to VGPR lanes (two v_readlane per use).  What is left is
// the CU's memory pipe: ~27 B/clk of stores, 256 KB (records) + 256 KB (fp32) per item.
// 32-bit lane offsets: 20 HW < 2^32 (fp32) and 32 planeO < 2^32 (records), checked on the host (rec_image_ok).
// ST (slow mode, round 5): the item's outputs also enter the GroupNorm statistics of the tensor (the producer of a POOLED norm's input,
// include/mdtile.h "Slow mode (round 5)"): a lane adds its NROW x NPX x 4 values of every 4-cout quad in fp32, the half-wave's 32 pixels
// are combined in fp64 and lane 0 of each half writes (sum, sum of squares) to E.st[(m * 8 + 2 g + kg) * 2 ..] -- combined over waves,
// items and cout quads in a fixed order by k_conv_stats_partial (vae_norm.hip).  Separate kernel symbols: the ST = false code is untouched.
template <int NPX, int NROW, int MW, int ECS = 64, bool ST = false>
__device__ __forceinline__ void epilogue_item(const EpiCtx& E, const u32x4* ec, f32x16 (&acc)[MW][NROW][NPX], int mt_local0, int mt_global0,
                                              const int (&ys)[NROW], int x, bool x_ok, const ResRows<NROW>& next) {
    constexpr bool ACC_RES = NPX == 1;     // the residual is already in the accumulators (see ResRows); the sub-pixel kernel keeps the plain form
    // The wave's MW 32-cout tiles are walked in UNITS of RB pixel rows: bias, (sub-pixel kernel: residual), fp32 stores, activation + split +
    // record stores, and -- one-pixel-per-lane kernels with a residual -- the same rows of the block's next item loaded into the registers
    // the unit has just stored from.
    constexpr int RB = NPX == 1 ? 2 : 1, UPM = NROW / RB, NU = MW * UPM;
    const size_t HW4 = E.HW * sizeof(float);
    const int xc = x_ok ? x : 0;                                  // clamped column for the unconditional residual loads
    unsigned kgo = (unsigned)E.kg;
    asm volatile("" : "+v"(kgo));      // what hangs on the lane's half (kg) is formed per item: as loop invariants these offsets end up in scratch
    const unsigned lane32 = ((4u * kgo) * (unsigned)E.HW + (unsigned)xc) * 4u;
    const int Pn = E.Cout >> 3;
    const size_t pl16 = E.planeO * sizeof(u32x4);
    const size_t lo_half = (size_t)Pn * pl16;
    const unsigned rlane = (kgo * (unsigned)E.planeO + (unsigned)(xc + mdt::REC_COL0)) * 16u;      // padded column 0 of ... + x

    float st1[ST ? MW : 1][4], st2[ST ? MW : 1][4];
    if constexpr (ST) {
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) st1[m][g] = st2[m][g] = 0.f;
    }
    float rbuf[RB][NPX][16];              // (the sub-pixel kernel's residual rows: read here, in the plain form -- the decoder never gives it one)
    auto request_residual = [&](int u, float (&r)[RB][NPX][16]) {
        const int m = u / UPM, n0 = (u % UPM) * RB;
        const char* rb = reinterpret_cast<const char*>(E.res) + ((size_t)E.b * E.Cout + (size_t)(mt_global0 + m) * 32) * HW4;
#pragma unroll
        for (int n = 0; n < RB; ++n) {
            const int yc = ys[n0 + n] < E.H ? ys[n0 + n] : E.H - 1;
            const unsigned ro = lane32 + (unsigned)(yc * E.W) * 4u;
            gchar* up = uniform_ptr(rb);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q) up = uniform_ptr(up + MDT_PLANE_STEP(q) * HW4);
                gchar* rp = up + (size_t)ro;
                if (NPX == 2) {
                    const f32x2 r2 = *(const MDT_GLOBAL f32x2*)rp;
                    r[n][0][q] = r2.x;
                    r[n][NPX - 1][q] = r2.y;
                } else {
                    r[n][0][q] = *(const MDT_GLOBAL float*)rp;
                }
            }
        }
    };

#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int m = u / UPM, n0 = (u % UPM) * RB;
        const int mt_global = mt_global0 + m;
        // constants of this lane's 16 couts of tile m: q = 4 g + i  <->  channel 32 mt + 4 kg + 8 g + i; read from LDS where they are
        // used (bias: once per tile; (a, s): per pixel row) -- kept in registers across a unit they were the first thing hipcc spilled
        const float4* e4 = reinterpret_cast<const float4*>(ec) + ((mt_local0 + m) * 8 + kgo);
        if (u % UPM == 0 && E.has_bias) {
            f32x2 bq[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 tb = e4[2 * g];
                bq[2 * g] = f32x2{tb.x, tb.y}; bq[2 * g + 1] = f32x2{tb.z, tb.w};
            }
#pragma unroll
            for (int n = 0; n < NROW; ++n)
#pragma unroll
                for (int e = 0; e < NPX; ++e)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        acc[m][n][e][2 * j] += bq[j].x;
                        acc[m][n][e][2 * j + 1] += bq[j].y;
                    }
        }
        if (!ACC_RES && E.res) {
            request_residual(u, rbuf);
#pragma unroll
            for (int n = 0; n < RB; ++n)
#pragma unroll
                for (int e = 0; e < NPX; ++e)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[m][n0 + n][e][q] += rbuf[n][e][q];
        }
        // ---- fp32 tensor: plane of value q = uniform base + ((q&3) + 8 (q>>2)) HW; lane = 4 kg HW + y W + x
        // ---- record image: plane ((mt*2 + R)*2 + kg) of the hi half, the lo half Cout/8 planes further; lane = kg plane + row + column
        const size_t slab = ((size_t)E.b * E.Cout + (size_t)mt_global * 32) * HW4;
        char* const yb32 = reinterpret_cast<char*>(E.y32) + slab;
        char* const yr = reinterpret_cast<char*>(E.yrec) + ((size_t)E.b * 2 * Pn + (size_t)mt_global * 4) * pl16;
        const bool whole = MW > 1 || mt_global * 32 + 32 <= E.Cout;   // narrow convs (conv_out: 3 couts; the one-tile-per-wave kernel only): couts past Cout are padding
#pragma unroll
        for (int nn = 0; nn < RB; ++nn) {
            const int n = n0 + nn;
            const int y = ys[n];
            if (!(y < E.H && x_ok)) continue;
            if constexpr (ST) {
#pragma unroll
                for (int e = 0; e < NPX; ++e)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float v = acc[m][n][e][q];
                        st1[m][q >> 2] += v;
                        st2[m][q >> 2] = fmaf(v, v, st2[m][q >> 2]);
                    }
            }
            if (E.y32 && !(E.dbg & 16)) {
                const unsigned ro = lane32 + (unsigned)(y * E.W) * 4u;
                if (whole) {
                    gchar* up = uniform_ptr(yb32);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        if (q) up = uniform_ptr(up + MDT_PLANE_STEP(q) * HW4);
                        gchar* yp = up + (size_t)ro;
                        if (NPX == 2) *(MDT_GLOBAL f32x2*)yp = f32x2{acc[m][n][0][q], acc[m][n][NPX - 1][q]};
                        else *(MDT_GLOBAL float*)yp = acc[m][n][0][q];
                    }
                } else {
                    const int cbase = mt_global * 32 + 4 * E.kg;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        if (cbase + (q & 3) + 8 * (q >> 2) >= E.Cout) continue;
                        char* yp = yb32 + (size_t)((q & 3) + 8 * (q >> 2)) * HW4 + (size_t)ro;
                        if (NPX == 2) *reinterpret_cast<float2*>(yp) = make_float2(acc[m][n][0][q], acc[m][n][NPX - 1][q]);
                        else *reinterpret_cast<float*>(yp) = acc[m][n][0][q];
                    }
                }
            }
            if (E.yrec) {
                const unsigned rrow = rlane + (unsigned)((y + 1) * E.WpO) * 16u;      // record (row y + 1, padded column x) of plane kg
#pragma unroll
                for (int R = 0; R < 2; ++R) {
                    if (E.has_act) {
                        f32x2 aq[4], sq[4];
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const float4 ta = e4[ECS + 4 * R + 2 * g], ts = e4[2 * ECS + 4 * R + 2 * g];
                            aq[2 * g] = f32x2{ta.x, ta.y}; aq[2 * g + 1] = f32x2{ta.z, ta.w};
                            sq[2 * g] = f32x2{ts.x, ts.y}; sq[2 * g + 1] = f32x2{ts.z, ts.w};
                        }
#pragma unroll
                        for (int e = 0; e < NPX; ++e) act8(acc[m][n][e], 8 * R, aq, sq);
                    }
                    gchar* const yp = uniform_ptr(yr + (size_t)(2 * R) * pl16), *const ypl = uniform_ptr(yp + lo_half);
#pragma unroll
                    for (int e = 0; e < NPX; ++e) {
                        u32x4 hi, lo;
                        split8p(acc[m][n][e], 8 * R, hi, lo);
                        const size_t at = (size_t)(rrow + (unsigned)(e + 1) * 16u);
                        if (!(E.dbg & 32)) {
                            *(MDT_GLOBAL u32x4*)(yp + at) = hi;
                            *(MDT_GLOBAL u32x4*)(ypl + at) = lo;
                        }
                    }
                }
                // zero border of the record image (this block owns the border cells next to its edge pixels)
                const bool left = x == 0, right = x + NPX == E.W, top = y == 0, bot = y == E.H - 1;
                if (left || right || top || bot) {
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    const unsigned p0 = (kgo * (unsigned)E.planeO + (unsigned)mdt::REC_COL0) * 16u;
#pragma unroll
                    for (int R = 0; R < 2; ++R) {
                        char* const yp = yr + (size_t)(2 * R) * pl16;
                        auto zrec = [&](int py, int px) {      // (px: padded column, 0 = left border)
                            const size_t at = (size_t)(p0 + (unsigned)(py * E.WpO + px) * 16u);
                            *reinterpret_cast<u32x4*>(yp + at) = z;
                            *reinterpret_cast<u32x4*>(yp + lo_half + at) = z;
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
        // the unit's accumulator rows are stored: the same rows of the block's next item (its residual) go into them now
        if constexpr (NPX == 1) {
            if (E.res && next.on) residual_into_acc<NROW, MW, RB>(E.res, E.Cout, E.HW, E.H, E.W, E.kg, next, m, n0, acc);
        }
    }
    if constexpr (ST) {
        const bool writer = (__lane_id() & 31) == 0;
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double d1 = (double)st1[m][g], d2 = (double)st2[m][g];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) {       // xor offsets < 32: inside the half-wave (the 32 pixels of a row)
                    d1 += __shfl_xor(d1, off, 64);
                    d2 += __shfl_xor(d2, off, 64);
                }
                if (writer) {
                    double* o = E.st + ((m * 8 + 2 * g) + E.kg) * 2;
                    o[0] = d1;
                    o[1] = d2;
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
