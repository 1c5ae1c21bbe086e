// 3x3 conv tiles of the Tiled-VAE task queue whose INPUT is already a split-bf16 "record image" written by the PRODUCER
// (the previous conv's epilogue, or k_rec_from_f32) -- fast-mode path of scripts/tilevae.py:507-656.
//
// Why a second conv family next to vae_conv_bf16x3.hip: in fast mode every GroupNorm's statistics are frozen BEFORE the
// tiles run (upstream tilevae.py:464-505, 542-563), so the (a, s) pair of the norm that FOLLOWS a conv is known when that
// conv's epilogue runs.  The epilogue therefore writes  silu(a * y + s)  already split into bf16 (hi, lo) halves in the
// order the next conv's MFMA fragments want.  The consumer then has NO producer work left: no fp32->bf16 split, no
// exp / rcp, no zero-pad mask -- its input tile goes HBM/L2 -> LDS by DMA (global_load_lds_dwordx4) exactly like the
// weights, and the main loop is ds_read_b128 + v_mfma only.  (round-1 counters: 1.3 VALU per MFMA and 4x fabric re-reads
// came from every cout block re-doing that producer work on the same pixels.)
//
// Record image of an activation [B, C, H, W], C % 32 == 0  (4 bytes per element, the same as fp32):
//     rec[b][hl][plane = C/8][H + 2][W + 2] x 16 B      hl = 0: bf16(x), hl = 1: bf16(x - hi)
//     one record = 8 channels of one pixel; a 1-pixel ZERO border is part of the image (the conv's zero padding and the
//     DMA's only "mask": out-of-image taps read border records, ragged block edges clamp onto it)
//     plane p = 2 * kstep + kg holds channels  32*(kstep>>1) + 16*(kstep&1) + 4*kg + (j&3) + 8*(j>>2),  j = 0..7
//     = the channels ONE LANE of the producing MFMA accumulator tile owns (C/D layout of v_mfma_f32_32x32x16: lane half kg
//     holds rows 4*kg + (q&3) + 8*(q>>2)), so the epilogue stores whole 16-byte records without any cross-lane movement;
//     the consumer's weights are packed with the same channel permutation inside each 16-channel K-step (pack time, free).
//
// Upstream call sites replaced: conv1 / conv2 / upsample.conv tasks of scripts/tilevae.py:115-195 together with the
// custom_group_norm + SiLU tasks in front of them (:218-245, :102-104), the queue's add_res (:614-616) and ldm's
// F.interpolate(nearest 2x) -- same set as vae_conv_bf16x3.hip, the arithmetic contract (split operands, 3 bf16 MFMAs per
// fp32-class product, fp32 accumulate) is identical.
//
// Kernel shape: block = 512 threads = 8 waves (2 per SIMD, 256-register budget, ONE block per CU), output tile
// BM couts x TH rows x 32 px with wave tile (32*MW couts) x NROW rows.  Main tile: 128 couts x 16 rows, wave 64 x 4 rows:
// 8 accumulator tiles, 72 MFMAs per wave and phase.  K loop = phases (16-channel K-step k, tap row dy), each split in three
// steps dx.  Operands: input stage [hl][kg][TH+2][34] records, 2 stages (K-step parity); weight chunks [hl][dx][mt][lane]
// in a 3-slot ring (slot = dy).  Software pipeline: the fragments of step t+1 are read from LDS BEFORE the MFMAs of step t
// are issued (two register sets), also across the phase boundary, so the matrix pipe never waits for an LDS round trip; the
// ONE block barrier of a phase sits after its dx = 0 step, where both waves of a SIMD still have two steps of MFMAs queued
// on either side; the DMA of weight chunk p+2 (and of the next K-step's input tile) is issued right behind that barrier
// and drained (vmcnt(0)) right before the next one -- a whole phase later.
#include "common.h"

using namespace mdt;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

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
    int ptiles, PX, NCB, NK;  // pixel tiles, tiles per row, cout blocks, 16-channel K-steps
};

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

// ---- shared epilogue: acc tile (32 couts x 32 px of one row) -> + bias (+ residual) -> fp32 NCHW and / or record image ----
// C/D layout of a 32x32 MFMA: col = lane & 31 (pixel), row = (q&3) + 8*(q>>2) + 4*(lane>>5) (cout).
struct EpiCtx {
    const ConvRParams* P;
    int b, kg;
    size_t HW, planeO;   // fp32 plane, record plane ((H+2)*(W+2))
    int WpO;
};

template <int NPX>   // NPX = 1: one pixel per lane; NPX = 2: the lane owns output px (2X, 2X+1) (sub-pixel upsample kernel)
__device__ __forceinline__ void epilogue_tile(const EpiCtx& E, const f32x16 (&acc)[NPX], int mt_global, int y, int x) {
    const ConvRParams& P = *E.P;
    const int cbase = mt_global * 32 + 4 * E.kg;
    float v[NPX][16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int co = cbase + (q & 3) + 8 * (q >> 2);
        const float bq = P.bias ? P.bias[co] : 0.0f;
#pragma unroll
        for (int e = 0; e < NPX; ++e) v[e][q] = acc[e][q] + bq;
    }
    const size_t o0 = ((size_t)E.b * P.Cout + cbase) * E.HW + (size_t)y * P.W + x;
    if (P.res) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float* rp = P.res + o0 + (size_t)((q & 3) + 8 * (q >> 2)) * E.HW;
            if (NPX == 2) {
                const float2 r2 = *reinterpret_cast<const float2*>(rp);
                v[0][q] += r2.x;
                v[NPX - 1][q] += r2.y;
            } else {
                v[0][q] += *rp;
            }
        }
    }
    if (P.y32) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float* yp = P.y32 + o0 + (size_t)((q & 3) + 8 * (q >> 2)) * E.HW;
            if (NPX == 2) *reinterpret_cast<float2*>(yp) = make_float2(v[0][q], v[NPX - 1][q]);
            else *yp = v[0][q];
        }
    }
    if (P.yrec) {
        if (P.coef) {
            const float* cf = P.coef + (size_t)E.b * 2 * P.Cout + cbase;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int cc = (q & 3) + 8 * (q >> 2);
                const float a = cf[cc], s = cf[P.Cout + cc];
#pragma unroll
                for (int e = 0; e < NPX; ++e) v[e][q] = silu_f(fmaf(v[e][q], a, s));
            }
        }
        // records R = 0 (q 0..7) and R = 1 (q 8..15) of this lane: planes ((mt*2 + R)*2 + kg)
        const int Pn = P.Cout >> 3;
        u32x4* yb = P.yrec + (size_t)E.b * 2 * Pn * E.planeO;
#pragma unroll
        for (int R = 0; R < 2; ++R) {
            const size_t pl = (size_t)((mt_global * 2 + R) * 2 + E.kg) * E.planeO;
#pragma unroll
            for (int e = 0; e < NPX; ++e) {
                float t8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t8[j] = v[e][8 * R + j];
                u32x4 hi, lo;
                split8r(t8, hi, lo);
                const size_t at = pl + (size_t)(y + 1) * E.WpO + (x + e + 1);
                yb[at] = hi;
                yb[(size_t)Pn * E.planeO + at] = lo;
            }
            // zero border of the record image (this block owns the border cells next to its edge pixels)
            const u32x4 z = {0u, 0u, 0u, 0u};
            auto zrec = [&](int py, int px) {
                const size_t at = pl + (size_t)py * E.WpO + px;
                yb[at] = z;
                yb[(size_t)Pn * E.planeO + at] = z;
            };
            const bool left = x == 0, right = x + NPX == P.W, top = y == 0, bot = y == P.H - 1;
            if (left) zrec(y + 1, 0);
            if (right) zrec(y + 1, P.W + 1);
            if (top) {
#pragma unroll
                for (int e = 0; e < NPX; ++e) zrec(0, x + e + 1);
                if (left) zrec(0, 0);
                if (right) zrec(0, P.W + 1);
            }
            if (bot) {
#pragma unroll
                for (int e = 0; e < NPX; ++e) zrec(P.H + 1, x + e + 1);
                if (left) zrec(P.H + 1, 0);
                if (right) zrec(P.H + 1, P.W + 1);
            }
        }
    }
}

#define MDT_PIN() __builtin_amdgcn_sched_barrier(0)

// =====================================================================================================================
// LDS input stage shared by both kernels: [hl][kg][ROWS][34] records, each hl half padded to whole 64-record DMA pieces so
// that one wave-instruction never straddles the two halves (hl then sits in the scalar base address, the lane offset
// stays 32-bit: global_load_lds with saddr + voffset).
template <int ROWS>
struct InStage {
    static constexpr int COLS = 34;
    static constexpr int HALF = 2 * ROWS * COLS;               // records of one hl half
    static constexpr int HALF_DMA = (HALF + 63) / 64;           // wave-instructions per half
    static constexpr int HALF_PAD = HALF_DMA * 64;
    static constexpr int DMA = 2 * HALF_DMA, PAD = 2 * HALF_PAD;
    static constexpr int PW = (DMA + 7) / 8;                    // wave-instructions per wave
};

// direct 3x3:  MW = 32-cout tiles per wave (2), WM = waves along cout, NROW = pixel rows per wave (4: two half-steps of 2)
template <int MW, int WM, int NROW>
__global__ __launch_bounds__(512, 2) void k_conv3x3_rec(const ConvRParams P) {
    constexpr int WR = 8 / WM, TH = WR * NROW, MT = MW * WM, HN = NROW / 2;
    constexpr int ROWS = TH + 2, COLS = 34;
    using IS = InStage<ROWS>;
    constexpr int W_REC = 2 * 3 * MT * 64;              // [hl][dx][mt][lane]
    constexpr int W_DMA = W_REC / 64;
    constexpr int W_PW = (W_DMA + 7) / 8;
    __shared__ u32x4 smem[2 * IS::PAD + 3 * W_REC];
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + 2 * IS::PAD;

    // block -> (pixel tile, cout block): XCD = id % 8 keeps all cout blocks of a pixel tile on one L2
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int ptile = (slot / P.NCB) * 8 + xcd, cb = slot % P.NCB;
    if (ptile >= P.ptiles) return;
    const int b = blockIdx.y;
    const int py = ptile / P.PX, px = ptile - py * P.PX;
    const int y0 = py * TH, x0 = px * 32;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wr = wave / WM;
    const int Hp = P.H + 2, Wp = P.W + 2, Pn = P.Cin >> 3;
    const size_t plane = (size_t)Hp * Wp;
    const char* xb = reinterpret_cast<const char*>(P.x + (size_t)b * 2 * Pn * plane);

    // input DMA map: wave-instruction di = wave + 8 i covers LDS records [64 di, 64 di + 64) of a stage; hl = di / HALF_DMA
    unsigned ioff[IS::PW];   // byte offset inside the (K-step, hl) pair of planes
#pragma unroll
    for (int i = 0; i < IS::PW; ++i) {
        const int di = wave + 8 * i;
        int s = (di % IS::HALF_DMA) * 64 + lane;
        if (s >= IS::HALF) s = IS::HALF - 1;            // pad lanes shadow the last record (they land in the pad area)
        const int g = s / (ROWS * COLS), p = s - g * (ROWS * COLS);
        const int r = p / COLS, c = p - r * COLS;
        int pr = y0 + r, pc = x0 + c;                   // padded coordinates (image row y0 + r - 1)
        pr = pr < Hp ? pr : Hp - 1;                     // ragged block edge: clamp onto the zero border
        pc = pc < Wp ? pc : Wp - 1;
        ioff[i] = (unsigned)(((size_t)g * plane + (size_t)pr * Wp + pc) * 16);
    }
    auto issue_input = [&](int k, int stage) {
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            const int di = wave + 8 * i;
            if (di < IS::DMA) {
                const char* base = xb + ((size_t)(di / IS::HALF_DMA) * Pn + 2 * (size_t)k) * plane * 16;   // wave-uniform
                dma16(base, ioff[i], in_l + stage * IS::PAD + di * 64);
            }
        }
    };
    const char* wsrc = reinterpret_cast<const char*>(P.w + (size_t)cb * P.NK * 3 * W_REC);
    const unsigned lane16 = lane * 16;
    auto issue_weights = [&](int ph, int ring) {
#pragma unroll
        for (int i = 0; i < W_PW; ++i)
            if (wave + 8 * i < W_DMA) {
                const char* base = wsrc + ((size_t)ph * W_REC + (wave + 8 * i) * 64) * 16;
                dma16(base, lane16, w_l + ring * W_REC + (wave + 8 * i) * 64);
            }
    };

    bf16x8 fw[2][MW][2];   // [set][m][hl]
    bf16x8 fx[2][HN][2];   // [set][row of the half-step][hl]
    const int wfrag = wm * MW * 64 + lane;                       // + ((hl*3 + dx)*MT + m)*64
    const int xfrag = (kg * ROWS + wr * NROW) * COLS + l31;      // + hl*HALF_PAD + (n + dy)*COLS + dx
    auto load_fw = [&](int set, int ring, int dx) {
        const u32x4* wst = w_l + ring * W_REC + wfrag;
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fw[set][m][hl] = __builtin_bit_cast(bf16x8, wst[((hl * 3 + dx) * MT + m) * 64]);
    };
    auto load_fx = [&](int set, int stage, int dy, int dx, int h) {
        const u32x4* ist = in_l + stage * IS::PAD + xfrag + (dy + h * HN) * COLS + dx;
#pragma unroll
        for (int n = 0; n < HN; ++n)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fx[set][n][hl] = __builtin_bit_cast(bf16x8, ist[hl * IS::HALF_PAD + n * COLS]);
    };

    f32x16 acc[MW][NROW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NROW; ++n)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.0f;

    const int nph = P.NK * 3;
    issue_input(0, 0);
    issue_weights(0, 0);
    issue_weights(1, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_fw(0, 0, 0);
    load_fx(0, 0, 0, 0, 0);

    // one trip = 2 K-steps = 6 phases = 18 steps = 36 half-steps: ring slot (= dy), input stage (= kk) and both register-set
    // parities are compile-time constants inside the unrolled body
    for (int k2 = 0; k2 < P.NK; k2 += 2) {
#pragma unroll
        for (int t = 0; t < 36; ++t) {
            const int kk = t / 18, dy = (t / 6) % 3, dx = (t / 2) % 3, h = t & 1;
            const int k = k2 + kk, ph = k * 3 + dy;
            const int xs = t & 1, ws = (t >> 1) & 1;
            // ---- the NEXT half-step's fragments go out first
            MDT_PIN();
            if (h == 0) {
                load_fx(xs ^ 1, kk, dy, dx, 1);
            } else if (t < 35) {
                const int t1 = t + 1, kk1 = t1 / 18, dy1 = (t1 / 6) % 3, dx1 = (t1 / 2) % 3;
                load_fw(ws ^ 1, dy1, dx1);
                load_fx(xs ^ 1, kk1, dy1, dx1, 0);
            } else if (k2 + 2 < P.NK) {
                load_fw(ws ^ 1, 0, 0);
                load_fx(xs ^ 1, 0, 0, 0, 0);
            }
            MDT_PIN();
            // ---- this half-step's MFMAs: term-major over its accumulators (a dependent MFMA is MW*HN issues away)
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int n = 0; n < HN; ++n)
#pragma unroll
                    for (int m = 0; m < MW; ++m)
                        acc[m][h * HN + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ws][m][term == 0 ? 1 : 0], fx[xs][n][term == 1 ? 1 : 0],
                                                                                     acc[m][h * HN + n], 0, 0, 0);   // w_lo x_hi, w_hi x_lo, w_hi x_hi
            MDT_PIN();
            if (dx == 0 && h == 1) {
                // chunk ph+1 (and, one phase after it was issued, the input tile of K-step k+1) went out behind the previous barrier
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share has landed
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // DMA issue of this phase (~10 scalar / VMEM instructions per piece), staggered between the two waves that share a
            // SIMD (w and w + 4): waves 0-3 right behind the barrier, waves 4-7 one half-step later -- while one of the pair
            // issues its pieces the other one keeps the matrix pipe fed.
            // Ring slot of chunk ph+2 = the one chunk ph-1 held: every wave finished reading it before the barrier.
            if ((dx == 0 && h == 1 && wave < 4) || (dx == 1 && h == 0 && wave >= 4)) {
                if (ph + 2 < nph) issue_weights(ph + 2, (dy + 2) % 3);
                if (dy == 0 && k + 1 < P.NK) issue_input(k + 1, (kk + 1) & 1);
            }
        }
    }

    EpiCtx E;
    E.P = &P; E.b = b; E.kg = kg;
    E.HW = (size_t)P.H * P.W; E.planeO = plane; E.WpO = Wp;
    const int x = x0 + l31;
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NROW; ++n) {
            const int y = y0 + wr * NROW + n;
            if (y < P.H && x < P.W) {
                const f32x16 a1[1] = {acc[m][n]};
                epilogue_tile<1>(E, a1, cb * MT + wm * MW + m, y, x);
            }
        }
}

// =====================================================================================================================
// nearest-2x upsample + 3x3 conv in sub-pixel form (four 2x2 convs on the un-upsampled grid, see vae_conv_bf16x3.hip
// k_upconv_bf16x3 for the derivation).  Block = 128 couts x (8 x 32 INPUT px) of ONE output-row parity a and both column
// parities bb; phases (K-step k, tap row u); a phase = 4 combo-steps c of 12 MFMAs per wave:
//     c:  0 (shift s 0, bb 0)   1 (s 1, bb 0)   2 (s 1, bb 1)   3 (s 2, bb 1)        tap column v = s - bb
// Weight chunk [hl][bb][v][mt][lane] per (a, cb, k, u); ring slot = phase % 3.
__global__ __launch_bounds__(512, 2) void k_upconv_rec(const ConvRParams P) {
    constexpr int MT = 4, MW = 2, WM = 2, NROW = 2, TH = 8;
    constexpr int ROWS = TH + 2, COLS = 34;
    using IS = InStage<ROWS>;
    constexpr int W_REC = 2 * 2 * 2 * MT * 64, W_DMA = W_REC / 64, W_PW = W_DMA / 8;
    __shared__ u32x4 smem[2 * IS::PAD + 3 * W_REC];
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + 2 * IS::PAD;

    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int per = P.NCB * 2;
    const int ptile = (slot / per) * 8 + xcd, rem = slot % per, cb = rem >> 1, a = rem & 1;
    if (ptile >= P.ptiles) return;
    const int b = blockIdx.y;
    const int py = ptile / P.PX, px = ptile - py * P.PX;
    const int y0 = py * TH, x0 = px * 32;              // INPUT coordinates

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wr = wave / WM;
    const int Hp = P.Hin + 2, Wp = P.Win + 2, Pn = P.Cin >> 3;
    const size_t plane = (size_t)Hp * Wp;
    const char* xb = reinterpret_cast<const char*>(P.x + (size_t)b * 2 * Pn * plane);

    unsigned ioff[IS::PW];
#pragma unroll
    for (int i = 0; i < IS::PW; ++i) {
        const int di = wave + 8 * i;
        int s = (di % IS::HALF_DMA) * 64 + lane;
        if (s >= IS::HALF) s = IS::HALF - 1;
        const int g = s / (ROWS * COLS), p = s - g * (ROWS * COLS);
        const int r = p / COLS, c = p - r * COLS;
        int pr = y0 + r, pc = x0 + c;
        pr = pr < Hp ? pr : Hp - 1;
        pc = pc < Wp ? pc : Wp - 1;
        ioff[i] = (unsigned)(((size_t)g * plane + (size_t)pr * Wp + pc) * 16);
    }
    auto issue_input = [&](int k, int stage) {
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            const int di = wave + 8 * i;
            if (di < IS::DMA) {
                const char* base = xb + ((size_t)(di / IS::HALF_DMA) * Pn + 2 * (size_t)k) * plane * 16;
                dma16(base, ioff[i], in_l + stage * IS::PAD + di * 64);
            }
        }
    };
    const int nph = P.NK * 2;
    const char* wsrc = reinterpret_cast<const char*>(P.w + ((size_t)a * P.NCB + cb) * nph * W_REC);
    const unsigned lane16 = lane * 16;
    auto issue_weights = [&](int ph, int ring) {
#pragma unroll
        for (int i = 0; i < W_PW; ++i) {
            const char* base = wsrc + ((size_t)ph * W_REC + (wave + 8 * i) * 64) * 16;
            dma16(base, lane16, w_l + ring * W_REC + (wave + 8 * i) * 64);
        }
    };

    bf16x8 fw[2][MW][2];     // [set][m][hl]   weight tiles of one combo-step
    bf16x8 fx[2][NROW][2];   // [set][n][hl]   input rows of one column shift
    const int wfrag = wm * MW * 64 + lane;
    const int xfrag = (kg * ROWS + wr * NROW + a) * COLS + l31;   // halo row of output row n at tap row u: + (n + u)*COLS
    auto load_fw = [&](int set, int ring, int c) {
        const int bb = c >> 1, v = ((c + 1) >> 1) - bb;           // c: 0 -> (0, 0), 1 -> (0, 1), 2 -> (1, 0), 3 -> (1, 1)
        const u32x4* wst = w_l + ring * W_REC + wfrag;
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fw[set][m][hl] = __builtin_bit_cast(bf16x8, wst[(((hl * 2 + bb) * 2 + v) * MT + m) * 64]);
    };
    auto load_fx = [&](int set, int stage, int u, int s) {
        const u32x4* ist = in_l + stage * IS::PAD + xfrag + u * COLS + s;
#pragma unroll
        for (int n = 0; n < NROW; ++n)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fx[set][n][hl] = __builtin_bit_cast(bf16x8, ist[hl * IS::HALF_PAD + n * COLS]);
    };

    f32x16 acc[2][MW][NROW];   // [bb][m][n]
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NROW; ++n)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[bb][m][n][q] = 0.0f;

    issue_input(0, 0);
    issue_weights(0, 0);
    issue_weights(1, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_fw(0, 0, 0);
    load_fx(0, 0, 0, 0);

    // one trip = 3 K-steps = 6 phases = 24 combo-steps: ring slot (phase % 3) and the register sets are compile-time.
    // fw set = combo-step parity; fx set = parity of the running shift counter 3*phase + s.  NK % 3 != 0: the surplus
    // K-steps of the last trip are skipped (wave-uniform branch).
    for (int k3 = 0; k3 < P.NK; k3 += 3) {
#pragma unroll
        for (int t = 0; t < 24; ++t) {
            const int pl_ = t >> 2, c = t & 3;                 // local phase 0..5, combo-step
            const int kk = pl_ >> 1, u = pl_ & 1, s = (c + 1) >> 1, bb = c >> 1;
            const int k = k3 + kk, ph = k * 2 + u;
            const int ws = t & 1, xs = (3 * pl_ + s) & 1;
            if (k < P.NK) {
                MDT_PIN();
                if (c < 3) {
                    load_fw(ws ^ 1, pl_ % 3, c + 1);
                    if (c != 1) load_fx(xs ^ 1, k & 1, u, s + 1);
                } else {
                    const int pl1 = (pl_ + 1) % 6, kk1 = pl1 >> 1, u1 = pl1 & 1;
                    const int k1 = (pl_ < 5 ? k3 : k3 + 3) + kk1;
                    if (k1 < P.NK) {
                        load_fw(ws ^ 1, pl1 % 3, 0);
                        load_fx(xs ^ 1, k1 & 1, u1, 0);
                    }
                }
                MDT_PIN();
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int n = 0; n < NROW; ++n)
#pragma unroll
                        for (int m = 0; m < MW; ++m)
                            acc[bb][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ws][m][term == 0 ? 1 : 0], fx[xs][n][term == 1 ? 1 : 0],
                                                                                    acc[bb][m][n], 0, 0, 0);
                MDT_PIN();
                if (c == 1) {
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                // staggered DMA issue (see k_conv3x3_rec): waves 0-3 behind the barrier, waves 4-7 one combo-step later
                if ((c == 1 && wave < 4) || (c == 2 && wave >= 4)) {
                    if (ph + 2 < nph) issue_weights(ph + 2, (pl_ + 2) % 3);
                    if (u == 0 && k + 1 < P.NK) issue_input(k + 1, (k + 1) & 1);
                }
            }
        }
    }

    EpiCtx E;
    E.P = &P; E.b = b; E.kg = kg;
    E.HW = (size_t)P.H * P.W; E.planeO = (size_t)(P.H + 2) * (P.W + 2); E.WpO = P.W + 2;
    const int xi = x0 + l31;
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NROW; ++n) {
            const int yi = y0 + wr * NROW + n;
            if (yi < P.Hin && xi < P.Win) {
                const f32x16 a2[2] = {acc[0][m][n], acc[1][m][n]};
                epilogue_tile<2>(E, a2, cb * MT + wm * MW + m, 2 * yi + a, 2 * xi);
            }
        }
}

// =====================================================================================================================
// fp32 NCHW -> record image (+ optional fixed-statistics GroupNorm + SiLU): entry points of the record path (conv_in /
// attention outputs, the fast-mode estimator and slow mode, where the statistics only exist after the producer ran).
// One thread = one (pixel, plane) of the PADDED image; border threads write the zero records.
__global__ __launch_bounds__(256) void k_rec_from_f32(const float* __restrict__ x, const float* __restrict__ coef, u32x4* __restrict__ rec,
                                                      int C, int H, int W) {
    const int Wp = W + 2, Hp = H + 2, Pn = C >> 3;
    const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y;
    const int bp = blockIdx.z, b = bp / Pn, p = bp - b * Pn;
    if (px >= Wp) return;
    const size_t planeO = (size_t)Hp * Wp;
    u32x4* hi_p = rec + ((size_t)b * 2 * Pn + p) * planeO + (size_t)py * Wp + px;
    u32x4* lo_p = hi_p + (size_t)Pn * planeO;
    u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
    if (px >= 1 && px <= W && py >= 1 && py <= H) {
        const int ks = p >> 1, g = p & 1;
        const int c0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * g;
        const float* src = x + ((size_t)b * C + c0) * H * W + (size_t)(py - 1) * W + (px - 1);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cc = (j & 3) + 8 * (j >> 2);
            float t = src[(size_t)cc * H * W];
            if (coef) t = silu_f(fmaf(t, coef[(size_t)b * 2 * C + c0 + cc], coef[(size_t)b * 2 * C + C + c0 + cc]));
            v[j] = t;
        }
        split8r(v, hi, lo);
    }
    *hi_p = hi;
    *lo_p = lo;
}

// record image -> fp32 NCHW (hi + lo): inspection / tests
__global__ __launch_bounds__(256) void k_rec_to_f32(const u32x4* __restrict__ rec, float* __restrict__ x, int C, int H, int W) {
    const int Wp = W + 2, Hp = H + 2, Pn = C >> 3;
    const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y;
    const int bp = blockIdx.z, b = bp / Pn, p = bp - b * Pn;
    if (px >= W) return;
    const size_t planeO = (size_t)Hp * Wp;
    const u32x4* hi_p = rec + ((size_t)b * 2 * Pn + p) * planeO + (size_t)(py + 1) * Wp + (px + 1);
    const bf16x8 h = __builtin_bit_cast(bf16x8, *hi_p), l = __builtin_bit_cast(bf16x8, hi_p[(size_t)Pn * planeO]);
    const int ks = p >> 1, g = p & 1;
    const int c0 = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * g;
    float* dst = x + ((size_t)b * C + c0) * H * W + (size_t)py * W + px;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[(size_t)((j & 3) + 8 * (j >> 2)) * H * W] = (float)h[j] + (float)l[j];
}

}  // namespace

namespace mdt {

size_t conv_bf16x3_direct_records(int cout, int cin);   // vae_conv_bf16x3.hip

bool conv_rec_supported(int cout, int cin, int ksize) { return ksize == 3 && cin % 32 == 0 && cout % 128 == 0; }

size_t rec_image_bytes(int B, int C, int H, int W) { return (size_t)B * C * (H + 2) * (W + 2) * 4; }

int rec_from_f32_launch(const float* d_x, const float* d_coef, void* d_rec, int B, int C, int H, int W, hipStream_t s) {
    dim3 grid(cdiv(W + 2, 256), H + 2, B * (C / 8));
    hipLaunchKernelGGL(k_rec_from_f32, grid, dim3(256), 0, s, d_x, d_coef, (u32x4*)d_rec, C, H, W);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

int rec_to_f32_launch(const void* d_rec, float* d_x, int B, int C, int H, int W, hipStream_t s) {
    dim3 grid(cdiv(W, 256), H, B * (C / 8));
    hipLaunchKernelGGL(k_rec_to_f32, grid, dim3(256), 0, s, (const u32x4*)d_rec, d_x, C, H, W);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

int conv_rec_launch(const void* d_xrec, const void* d_w_rec, const float* d_bias, const float* d_res, float* d_y32, void* d_yrec,
                    const float* d_ycoef, int B, int cin, int cout, int H, int W, int up, hipStream_t s) {
    ConvRParams P;
    P.x = (const u32x4*)d_xrec; P.w = (const u32x4*)d_w_rec; P.bias = d_bias; P.res = d_res; P.y32 = d_y32;
    P.yrec = (u32x4*)d_yrec; P.coef = d_ycoef;
    P.B = B; P.Cin = cin; P.Cout = cout; P.H = H; P.W = W;
    P.Hin = up ? H / 2 : H; P.Win = up ? W / 2 : W;
    P.NCB = cout / 128;
    P.NK = cin / 16;
    if (up) {
        P.w = (const u32x4*)d_w_rec + conv_bf16x3_direct_records(cout, cin);
        P.PX = (P.Win + 31) / 32;
        P.ptiles = P.PX * ((P.Hin + 7) / 8);
        dim3 grid(((P.ptiles + 7) / 8) * 8 * P.NCB * 2, B), block(512);
        hipLaunchKernelGGL(k_upconv_rec, grid, block, 0, s, P);
        MDT_LAUNCH_CHECK();
        return MDTILE_OK;
    }
    P.PX = (W + 31) / 32;
    P.ptiles = P.PX * ((H + 15) / 16);
    dim3 grid(((P.ptiles + 7) / 8) * 8 * P.NCB, B), block(512);
    hipLaunchKernelGGL((k_conv3x3_rec<2, 2, 4>), grid, block, 0, s, P);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

}  // namespace mdt
