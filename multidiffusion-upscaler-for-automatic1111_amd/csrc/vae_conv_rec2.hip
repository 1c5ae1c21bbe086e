// Record-image 3x3 conv and sub-pixel upsample conv, TWO INDEPENDENT 4-WAVE BLOCKS PER CU (round 4).
//
// Same arithmetic, same record images, same packed weights and the same per-accumulator MFMA order as vae_conv_rec.hip (the
// results are bit-identical to that file's kernels); what changes is who shares a SIMD.  There one 512-thread block owns the CU:
// its two waves per SIMD run in lock step, and the item's store epilogue (12.8 us of an 87 us item record -> record, 33 us of
// 108 us with the fp32 residual / fp32 output streams -- 17 % of the family's time, profiles/r3g) runs with the matrix pipes
// idle, because vmcnt retires loads and stores in order: a wave cannot leave its stores draining behind the next item's DMA
// waits.  Only ANOTHER wave's MFMAs can cover them.  gfx950 has one barrier per workgroup, so that other wave has to live in
// another workgroup: here a CU holds two 256-thread blocks (one wave each per SIMD, <= 80 KB of LDS, <= 256 registers), each
// working through its own items, and the second block to arrive on a CU starts half an item late, so that one block's epilogue
// (and every barrier / DMA wait of its K loop) sits under the other block's K loop.
//
// What that costs and how it is paid:
//   * half the LDS per block: an item is 128 couts x 8 rows x 32 px (wave tile unchanged: 64 couts x 4 rows, 8 accumulator
//     tiles); the weight ring holds STEP chunks -- one (K-step, dy, dx) = [hl][mt][lane] = 8 KB -- in 4 slots instead of phase
//     chunks in 3, the input stage [hl][kg][10][34] records stays double-buffered: 45 + 32 + 3 KB = 79 KB;
//   * one block barrier per step (24 MFMAs per wave) instead of per phase (72): a wave that waits leaves its SIMD to the other
//     block's wave, which is the point;
//   * the weight stream from L2 doubles per MFMA (8-row items): 14 B/clk per CU, inside the L2s' rate.
// DMA protocol (global_load_lds_dwordx4 from inline asm, completion counted by hand, as in vae_conv_rec.hip): the chunk of step
// t+3 is requested right behind the barrier of step t (its slot held step t-1, which every wave has finished reading by then)
// and has to have landed at the barrier of step t+2; the input stage of K-step k+1 goes out one piece per wave and step during
// the first six steps of K-step k.  Every barrier is preceded by a COUNTED vmcnt: the pieces a wave has requested after the chunk
// the barrier publishes may stay in flight (N(s) below; waves that requested an extra piece -- the ragged sixth input piece, the
// epilogue constants -- merely wait for it too).  The stream of chunks / input stages runs on across item boundaries: the last
// K-step of an item requests the first operands of the block's next item.
//
// Upstream call sites replaced: the same as vae_conv_rec.hip (scripts/tilevae.py:115-195 conv1 / conv2 / upsample.conv tasks,
// :218-245 custom_group_norm, :102-104 SiLU, :614-616 add_res).
#include "common.h"

#include <mutex>

using namespace mdt;

#include "conv_rec_common.h"

namespace {

constexpr int NWV = 4;            // waves per block: one per SIMD; the CU's other four wave slots belong to the second block
constexpr int EC2 = 3 * 32;       // records of one constants buffer: [bias | a | s] x 512 B (requested by 32 lanes each)

#define MDT_WAITV(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))      // s_waitcnt vmcnt(n), n <= 15 (expcnt / lgkmcnt untouched)
#define MDT_BARRIER()                    \
    do {                                 \
        asm volatile("" ::: "memory");   \
        __builtin_amdgcn_s_barrier();    \
        asm volatile("" ::: "memory");   \
    } while (0)

struct Item2 {
    int b, cb, y0, x0;
};

// The block that arrives second on its CU waits `skew_ticks` before its first item (see the file header).  "Second" is decided
// by an arrival counter per hardware CU (XCC_ID, and SE / SH / CU id of HW_ID); without a counter buffer by the block index
// (blocks >= grid / 2 are dispatched after every CU got its first block).  Wave 0 only: the others wait at the item's first barrier.
__device__ __forceinline__ void startup_skew(const ConvRParams& P, int wave, int lane) {
    if (wave != 0) return;                                                   // (wave-uniform: the waits below are scalar loops)
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);           // HW_REG_HW_ID: [11:8] CU, [12] SH, [15:13] SE
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;    // HW_REG_XCC_ID[3:0]
    const unsigned key = (xcc << 8) | ((hw >> 8) & 255u);
    unsigned second = blockIdx.x >= gridDim.x / 2 ? 1u : 0u;
    if (P.cu_ctr) {
        // arrival number of this block on its CU IN THIS LAUNCH: the counter word carries the launch epoch, so a launch that left an odd
        // number of blocks on some CUs (grid < 2 CUs is common for this family) cannot flip the parity of every later launch there
        unsigned n = 0;
        if (lane == 0) {
            unsigned* c = P.cu_ctr + key;
            unsigned seen = *reinterpret_cast<volatile unsigned*>(c);
            while (true) {
                const unsigned cnt = (seen >> 8) == P.epoch ? (seen & 255u) : 0u;
                const unsigned prev = atomicCAS(c, seen, (P.epoch << 8) | ((cnt + 1u) & 255u));
                if (prev == seen) { n = cnt; break; }
                seen = prev;
            }
        }
        second = (unsigned)__builtin_amdgcn_readfirstlane((int)n) & 1u;
    }
    if (P.census && lane == 0) P.census[blockIdx.x] = key | (second << 31);
    if (second && P.skew_ticks) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < P.skew_ticks) __builtin_amdgcn_s_sleep(32);
    }
}

// =====================================================================================================================
// direct 3x3.  MW = 32-cout tiles per wave (2), WM = waves along cout (2), NROW = pixel rows per wave (4: two half-steps of 2).
// Step t = (k, dy, dx) of an item, 9 NK steps; chunk(t) sits in ring slot (r0 + t) & 3, r0 = the item's ring origin (the ring
// runs on across items: r0' = (r0 + 9 NK) & 3); K-step k uses input stage k & 1 (NK is even: every item starts in stage 0).
// One step of a wave:   fx(t, h=1) <- LDS | 12 MFMAs (rows 0-1) | vmcnt(N) + barrier | DMA requests |
//                       fw(t+1), fx(t+1, h=0) <- LDS | 12 MFMAs (rows 2-3)
template <int MW, int WM, int NROW>
__global__ __launch_bounds__(256, 2) void k_conv3x3_rec2(const ConvRParams P) {
    constexpr int WR = NWV / WM, TH = WR * NROW, MT = MW * WM, HN = NROW / 2;
    constexpr int ROWS = TH + 2, COLS = 34;
    using IS = InStage<ROWS, NWV>;
    constexpr int W_STEP = 2 * MT * 64;               // records of a step chunk [hl][mt][lane]
    constexpr int W_PH = 3 * W_STEP;                  // records of a packed phase chunk [hl][dx][mt][lane] (the layout in HBM)
    constexpr int W_PW = W_STEP / 64 / NWV;           // pieces per wave and step
    static_assert(W_STEP / 64 == W_PW * NWV && W_PW == 2, "a step chunk is two pieces per wave");
    static_assert(IS::PW == 6 && IS::DMA > 5 * NWV, "the counted waits below assume five input pieces from every wave (a sixth from some)");
    __shared__ u32x4 smem[2 * IS::PAD + 4 * W_STEP + 2 * EC2];
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + 2 * IS::PAD;
    u32x4* const ec_l = smem + 2 * IS::PAD + 4 * W_STEP;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wr = wave / WM;
    const int Hp = P.H + 2, Wp = rec_pitch(P.W), Pn = P.Cin >> 3;
    const size_t plane = (size_t)Hp * Wp;

    // work -> (sample, pixel tile, cout block): as in k_conv3x3_rec (grid % 8 == 0 => `work % 8` is this block's XCD for all its
    // items: the cout blocks of a pixel tile stay on one L2)
    const int per_img = ((P.ptiles + 7) / 8) * 8 * P.NCB, total = per_img * P.B;
    auto decode = [&](int work, Item2& it) -> bool {
        it.b = work / per_img;
        const int r = work - it.b * per_img, xcd = r & 7, slot = r >> 3;
        const int ptile = (slot / P.NCB) * 8 + xcd;
        it.cb = slot % P.NCB;
        const int py = ptile / P.PX, px = ptile - py * P.PX;
        it.y0 = py * TH;
        it.x0 = px * 32;
        return ptile < P.ptiles;
    };
    auto next_valid = [&](int work, Item2& it) -> int {
        while (work < total && !decode(work, it)) work += gridDim.x;
        return work;
    };

    // input DMA map: piece di = wave + 4 i covers LDS records [64 di, 64 di + 64) of a stage; hl = di / HALF_DMA
    auto make_ioff = [&](const Item2& it, unsigned (&ioff)[IS::PW]) {
        int ln = lane;
        asm volatile("" : "+v"(ln));      // (g, r, c) re-derived per item, as in k_conv3x3_rec: kept across the loop they go to scratch
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            const int di = wave + NWV * i;
            int s = (di % IS::HALF_DMA) * 64 + ln;
            if (s >= IS::HALF) s = IS::HALF - 1;            // pad lanes shadow the last record (they land in the pad area)
            const int g = s / (ROWS * COLS), p = s - g * (ROWS * COLS);
            const int r = p / COLS, c = p - r * COLS;
            int pr = it.y0 + r, pc = it.x0 + c;             // padded coordinates (image row y0 + r - 1, image column x0 + c - 1)
            pr = pr < Hp ? pr : Hp - 1;                     // ragged block edge: clamp onto the zero border
            pc = (pc < P.W + 1 ? pc : P.W + 1) + REC_COL0;  // (column of the record image: the left border sits at REC_COL0)
            ioff[i] = (unsigned)(((size_t)g * plane + (size_t)pr * Wp + pc) * 16);
        }
    };
    auto issue_input_piece = [&](const Item2& it, const unsigned (&ioff)[IS::PW], int k, int stage, int i) {
        const int di = wave + NWV * i;
        if (di < IS::DMA) {
            const char* xb = reinterpret_cast<const char*>(P.x + (size_t)it.b * 2 * Pn * plane);
            const char* base = xb + ((size_t)(di / IS::HALF_DMA) * Pn + 2 * (size_t)k) * plane * 16;   // wave-uniform
            dma16(base, ioff[i], in_l + stage * IS::PAD + di * 64);
        }
    };
    const unsigned lane16 = lane * 16;
    // step chunk (k, dy, dx) of the item's cout block -> ring slot: piece p = wave + 4 i = (hl, m-tile) = (i, wave)
    auto issue_wstep = [&](const Item2& it, int k, int dy, int dx, int slot) {
        const char* wsrc = reinterpret_cast<const char*>(P.w + ((size_t)it.cb * P.NK + k) * 3 * W_PH + (size_t)dy * W_PH);
#pragma unroll
        for (int i = 0; i < W_PW; ++i) {
            const int p = wave + NWV * i, hl = p / MT, j = p % MT;
            dma16(wsrc + (size_t)(((hl * 3 + dx) * MT + j) * 64) * 16, lane16, w_l + slot * W_STEP + p * 64);
        }
    };
    // epilogue constants of an item's 128 couts: waves 0 / 1 / 2 fetch bias / a / s, 512 B each (lanes 0-31)
    auto issue_consts = [&](const Item2& it, int par) {
        if (lane < 32) {
            if (wave == 0 && P.bias) dma16(reinterpret_cast<const char*>(P.bias + it.cb * (MT * 32)), lane16, ec_l + par * EC2);
            if ((wave == 1 || wave == 2) && P.yrec && P.coef)
                dma16(reinterpret_cast<const char*>(P.coef + ((size_t)it.b * 2 + (wave - 1)) * P.Cout + it.cb * (MT * 32)), lane16,
                      ec_l + par * EC2 + wave * 32);
        }
    };

    bf16x8 fw[2][MW][2];   // [set][m][hl]
    bf16x8 fx[2][HN][2];   // [set = half-step][row][hl]
    const int wfrag = wm * MW * 64 + lane;                       // + slot*W_STEP + (hl*MT + m)*64
    const int xfrag = (kg * ROWS + wr * NROW) * COLS + l31;      // + stage*PAD + hl*HALF_PAD + (n + dy)*COLS + dx
    auto load_fw = [&](int set, int slot) {
        const u32x4* wst = w_l + slot * W_STEP + wfrag;
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fw[set][m][hl] = __builtin_bit_cast(bf16x8, wst[(hl * MT + m) * 64]);
    };
    auto load_fx = [&](int set, int stage, int dy, int dx, int h) {
        const u32x4* ist = in_l + stage * IS::PAD + xfrag + (dy + h * HN) * COLS + dx;
#pragma unroll
        for (int n = 0; n < HN; ++n)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fx[set][n][hl] = __builtin_bit_cast(bf16x8, ist[hl * IS::HALF_PAD + n * COLS]);
    };

    Item2 cur, nxt;
    int work = next_valid(blockIdx.x, cur);
    if (work >= total) return;
    unsigned ioff[IS::PW];
    make_ioff(cur, ioff);
#pragma unroll
    for (int i = 0; i < IS::PW; ++i) issue_input_piece(cur, ioff, 0, 0, i);
    issue_wstep(cur, 0, 0, 0, 0);
    issue_wstep(cur, 0, 0, 1, 1);
    issue_wstep(cur, 0, 0, 2, 2);
    issue_consts(cur, 0);
    startup_skew(P, wave, lane);
    int par = 0, r0 = 0;

    // a conv2's residual arrives in the accumulators, as in k_conv3x3_rec (conv_rec_common.h: ResRows)
    f32x16 acc[MW][NROW][1];
    const bool res_in_acc = P.res != nullptr && !(pdbg(P.dbg) & 1);
    auto res_rows = [&](const Item2& it, bool on) {
        ResRows<NROW> R;
        R.on = on; R.b = it.b; R.mt_global0 = it.cb * MT + wm * MW;
#pragma unroll
        for (int n = 0; n < NROW; ++n) R.ys[n] = it.y0 + wr * NROW + n;
        int le = lane;
        asm volatile("" : "+v"(le));
        R.x = it.x0 + (le & 31);
        R.x_ok = R.x < P.W;
        return R;
    };
    if (res_in_acc) {
        const ResRows<NROW> R0 = res_rows(cur, true);
#pragma unroll
        for (int m = 0; m < MW; ++m) residual_into_acc<NROW, MW, NROW>(P.res, P.Cout, (size_t)P.H * P.W, P.H, P.W, kg, R0, m, 0, acc);
    }
    while (true) {
        MDT_WAITV(0);          // this wave's pieces of the item's first operands have landed (and its stores of the last item are out)
        MDT_BARRIER();
        load_fw(0, r0);
        load_fx(0, 0, 0, 0, 0);
        const int work_n = next_valid(work + gridDim.x, nxt);
        const bool has_next = work_n < total;
        unsigned ioff_n[IS::PW];
        if (has_next) make_ioff(nxt, ioff_n);

        if (!res_in_acc) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NROW; ++n)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[m][n][0][q] = 0.0f;
        }

        // one trip = 2 K-steps = 18 steps: register sets (fw: step parity, fx: half-step) and the input stage are compile-time,
        // the ring slot is rb + u (mod 4) with the trip's origin rb in a scalar register
        for (int k2 = 0; k2 < P.NK; k2 += 2) {
            const int rb = (r0 + k2) & 3;                     // 9 k2 = k2 (mod 4)
            const bool last_trip = k2 + 2 >= P.NK;
#pragma unroll
            for (int u = 0; u < 18; ++u) {
                const int kk = u / 9, s = u % 9, dy = s / 3, dx = s % 3;
                const int k = k2 + kk;
                const int ws = u & 1;
                // ---- half-step 0: rows 0 .. HN-1; the fragments of half-step 1 go out first
                MDT_PIN();
                load_fx(1, kk, dy, dx, 1);
                MDT_PIN();
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int n = 0; n < HN; ++n)
#pragma unroll
                        for (int m = 0; m < MW; ++m)
                            acc[m][n][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ws][m][term == 0 ? 1 : 0], fx[0][n][term == 1 ? 1 : 0],
                                                                                   acc[m][n][0], 0, 0, 0);   // w_lo x_hi, w_hi x_lo, w_hi x_hi
                MDT_PIN();
                // ---- the barrier of step t publishes chunk t+1 (and, at s = 8, the input stage of the next K-step).  Requested by
                // this wave after chunk t+1, oldest first: the input piece of step t-2, chunk t+2 (2 pieces), the input piece of
                // step t-1 -- input pieces go out at s = 0 .. 4 from every wave (s = 5: waves 0 / 1 only):
                //     s:  0  1  2  3  4  5  6  7  8
                //     N:  2  3  4  4  4  4  3  2  2
                // The last item of a block requests nothing in its last K-step: vmcnt(0).
                const bool tail = kk == 1 && last_trip && !has_next;
                if (tail) {
                    MDT_WAITV(0);
                } else if (s == 0 || s >= 7) {
                    MDT_WAITV(2);
                } else if (s == 1 || s == 6) {
                    MDT_WAITV(3);
                } else {
                    MDT_WAITV(4);
                }
                MDT_BARRIER();
                // ---- requests of step t: chunk t+3 into the slot of chunk t-1, one piece of the next K-step's input stage
                {
                    const int s3 = s + 3, slot3 = (rb + u + 3) & 3;
                    const bool into_next_item = kk == 1 && last_trip;      // "K-step k+1" is K-step 0 of the block's next item
                    if (s3 < 9) {
                        issue_wstep(cur, k, s3 / 3, s3 % 3, slot3);
                    } else if (!into_next_item) {
                        issue_wstep(cur, k + 1, (s3 - 9) / 3, (s3 - 9) % 3, slot3);
                    } else if (has_next) {
                        issue_wstep(nxt, 0, (s3 - 9) / 3, (s3 - 9) % 3, slot3);
                    }
                    if (s < IS::PW) {
                        if (!into_next_item) issue_input_piece(cur, ioff, k + 1, (kk + 1) & 1, s);
                        else if (has_next) issue_input_piece(nxt, ioff_n, 0, 0, s);
                    }
                    if (s == 6 && into_next_item && has_next) issue_consts(nxt, par ^ 1);
                }
                // ---- half-step 1: rows HN .. NROW-1; the fragments of step t+1 go out first
                MDT_PIN();
                if (u < 17) {
                    const int u1 = u + 1, kk1 = u1 / 9, s1 = u1 % 9;
                    load_fw(ws ^ 1, (rb + u1) & 3);
                    load_fx(0, kk1, s1 / 3, s1 % 3, 0);
                } else if (!last_trip) {
                    load_fw(ws ^ 1, (rb + 18) & 3);
                    load_fx(0, 0, 0, 0, 0);
                }
                MDT_PIN();
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int n = 0; n < HN; ++n)
#pragma unroll
                        for (int m = 0; m < MW; ++m)
                            acc[m][HN + n][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ws][m][term == 0 ? 1 : 0], fx[1][n][term == 1 ? 1 : 0],
                                                                                        acc[m][HN + n][0], 0, 0, 0);
                MDT_PIN();
            }
        }

        EpiCtx E;
        E.res = P.res; E.y32 = P.y32; E.yrec = P.yrec;
        E.has_bias = P.bias != nullptr; E.has_act = P.yrec != nullptr && P.coef != nullptr;
        E.Cout = P.Cout; E.H = P.H; E.W = P.W; E.b = cur.b; E.kg = kg;
        E.HW = (size_t)P.H * P.W; E.planeO = plane; E.WpO = Wp; E.dbg = pdbg(P.dbg);
        int le = lane;
        asm volatile("" : "+v"(le));      // (re-derived: a separate l31 kept alive through the epilogue goes to scratch)
        const int x = cur.x0 + (le & 31);
        int ys[NROW];
#pragma unroll
        for (int n = 0; n < NROW; ++n) ys[n] = cur.y0 + wr * NROW + n;
        if (!(pdbg(P.dbg) & 1)) {
            epilogue_item<1, NROW, MW, 32>(E, ec_l + par * EC2, acc, wm * MW, cur.cb * MT + wm * MW, ys, x, x < P.W, res_rows(nxt, has_next));
        }
        if (!has_next) break;
        work = work_n;
        cur = nxt;
        par ^= 1;
        r0 = (r0 + P.NK) & 3;                                // 9 NK = NK (mod 4)
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) ioff[i] = ioff_n[i];
    }
}

// =====================================================================================================================
// nearest-2x upsample + 3x3 conv in sub-pixel form (vae_conv_rec.hip: k_upconv_rec; derivation in vae_conv_bf16x3.hip), two blocks
// per CU.  Item = 128 couts x (4 x 32 INPUT px) of ONE output-row parity a and both column parities bb (wave tile 64 couts x 2
// input rows x 2 bb = 8 accumulator tiles).  Step t = (k, u, c): tap row u, combo-step c
//     c:  0 (shift s 0, bb 0)   1 (s 1, bb 0)   2 (s 1, bb 1)   3 (s 2, bb 1)        tap column v = s - bb
// 12 MFMAs per wave and step, 8 steps per K-step.  Step chunk = [hl][mt][lane] of (k, u, bb, v), 8 KB; the ring has SIX slots
// (the steps are half as long as the direct kernel's, so the same time in flight needs twice the chunks): the chunk of step t+5
// is requested behind the barrier of step t and has to have landed at the barrier of step t+4.  Input stage [hl][kg][6][34]
// records, double-buffered, 3-4 pieces per wave requested at the first steps of the previous K-step (before that step's chunk:
// the stage must be older than the chunk whose barrier publishes it).  28 + 48 + 3 KB = 79 KB.
// One step of a wave:   4 MFMAs (term 0) | vmcnt(N) + barrier | DMA requests | fw(t+1), fx(t+1) <- LDS | 8 MFMAs (terms 1, 2)
__device__ __forceinline__ int wrap6(int x) { return x >= 6 ? x - 6 : x; }

__global__ __launch_bounds__(256, 2) void k_upconv_rec2(const ConvRParams P) {
    constexpr int MT = 4, MW = 2, WM = 2, NROW = 2, TH = 4, R = 6;
    constexpr int ROWS = TH + 2, COLS = 34;
    using IS = InStage<ROWS, NWV>;
    constexpr int W_STEP = 2 * MT * 64;               // records of a step chunk [hl][mt][lane]
    constexpr int W_PH = 4 * W_STEP;                  // records of a packed phase chunk [hl][bb][v][mt][lane] (the layout in HBM)
    constexpr int W_PW = W_STEP / 64 / NWV;
    static_assert(W_PW == 2, "a step chunk is two pieces per wave");
    static_assert(IS::PW == 4 && IS::DMA > 3 * NWV, "the counted waits below assume three input pieces from every wave (a fourth from some)");
    __shared__ u32x4 smem[2 * IS::PAD + R * W_STEP + 2 * EC2];
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + 2 * IS::PAD;
    u32x4* const ec_l = smem + 2 * IS::PAD + R * W_STEP;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wr = wave / WM;
    const int Hp = P.HinF + 2, Wp = rec_pitch(P.WinF), Pn = P.Cin >> 3;      // pitches of the WHOLE input image; items tile its window
    const size_t plane = (size_t)Hp * Wp;

    struct Item {
        int b, cb, a, y0, x0;   // y0, x0: INPUT coordinates (relative to the window)
    };
    const int per = P.NCB * 2, per_img = ((P.ptiles + 7) / 8) * 8 * per, total = per_img * P.B;
    auto decode = [&](int work, Item& it) -> bool {
        it.b = work / per_img;
        const int r = work - it.b * per_img, xcd = r & 7, slot = r >> 3;
        const int ptile = (slot / per) * 8 + xcd, rem = slot % per;
        it.cb = rem >> 1;
        it.a = rem & 1;
        const int py = ptile / P.PX, px = ptile - py * P.PX;
        it.y0 = py * TH;
        it.x0 = px * 32;
        return ptile < P.ptiles;
    };
    auto next_valid = [&](int work, Item& it) -> int {
        while (work < total && !decode(work, it)) work += gridDim.x;
        return work;
    };
    auto make_ioff = [&](const Item& it, unsigned (&ioff)[IS::PW]) {
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            const int di = wave + NWV * i;
            int s = (di % IS::HALF_DMA) * 64 + lane;
            if (s >= IS::HALF) s = IS::HALF - 1;
            const int g = s / (ROWS * COLS), p = s - g * (ROWS * COLS);
            const int r = p / COLS, c = p - r * COLS;
            int pr = P.iy0[it.b & (REC_WIN_MAXB - 1)] + it.y0 + r, pc = P.ix0[it.b & (REC_WIN_MAXB - 1)] + it.x0 + c;   // inside the window's own border: the image's real neighbours
            pr = pr < Hp ? pr : Hp - 1;
            pc = (pc < P.WinF + 1 ? pc : P.WinF + 1) + REC_COL0;
            ioff[i] = (unsigned)(((size_t)g * plane + (size_t)pr * Wp + pc) * 16);
        }
    };
    auto issue_input_piece = [&](const Item& it, const unsigned (&ioff)[IS::PW], int k, int stage, int i) {
        const int di = wave + NWV * i;
        if (di < IS::DMA) {
            const char* xb = reinterpret_cast<const char*>(P.x + (size_t)it.b * 2 * Pn * plane);
            const char* base = xb + ((size_t)(di / IS::HALF_DMA) * Pn + 2 * (size_t)k) * plane * 16;
            dma16(base, ioff[i], in_l + stage * IS::PAD + di * 64);
        }
    };
    const int nph = P.NK * 2;
    const unsigned lane16 = lane * 16;
    // step chunk (k, u, c) of the item's (row parity, cout block) -> ring slot: piece p = wave + 4 i = (hl, m-tile) = (i, wave)
    auto issue_wstep = [&](const Item& it, int k, int u, int c, int slot) {
        const int bb = c >> 1, v = ((c + 1) >> 1) - bb;
        const char* wsrc = reinterpret_cast<const char*>(P.w + (((size_t)it.a * P.NCB + it.cb) * nph + (size_t)(k * 2 + u)) * W_PH);
#pragma unroll
        for (int i = 0; i < W_PW; ++i) {
            const int p = wave + NWV * i, hl = p / MT, j = p % MT;
            dma16(wsrc + (size_t)((((hl * 2 + bb) * 2 + v) * MT + j) * 64) * 16, lane16, w_l + slot * W_STEP + p * 64);
        }
    };
    auto issue_consts = [&](const Item& it, int par) {
        if (lane < 32) {
            if (wave == 0 && P.bias) dma16(reinterpret_cast<const char*>(P.bias + it.cb * (MT * 32)), lane16, ec_l + par * EC2);
            if ((wave == 1 || wave == 2) && P.yrec && P.coef)
                dma16(reinterpret_cast<const char*>(P.coef + ((size_t)it.b * 2 + (wave - 1)) * P.Cout + it.cb * (MT * 32)), lane16,
                      ec_l + par * EC2 + wave * 32);
        }
    };

    bf16x8 fw[2][MW][2];     // [set][m][hl]   weight tiles of one combo-step
    bf16x8 fx[2][NROW][2];   // [set][n][hl]   input rows of one column shift
    const int wfrag = wm * MW * 64 + lane;
    auto load_fw = [&](int set, int slot) {
        const u32x4* wst = w_l + slot * W_STEP + wfrag;
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fw[set][m][hl] = __builtin_bit_cast(bf16x8, wst[(hl * MT + m) * 64]);
    };
    auto load_fx = [&](int set, int xfrag, int stage, int u, int s) {
        const u32x4* ist = in_l + stage * IS::PAD + xfrag + u * COLS + s;
#pragma unroll
        for (int n = 0; n < NROW; ++n)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fx[set][n][hl] = __builtin_bit_cast(bf16x8, ist[hl * IS::HALF_PAD + n * COLS]);
    };

    Item cur, nxt;
    int work = next_valid(blockIdx.x, cur);
    if (work >= total) return;
    unsigned ioff[IS::PW];
    make_ioff(cur, ioff);
#pragma unroll
    for (int i = 0; i < IS::PW; ++i) issue_input_piece(cur, ioff, 0, 0, i);
#pragma unroll
    for (int t = 0; t < R - 1; ++t) issue_wstep(cur, t / 8, (t % 8) / 4, t % 4, t);      // (NK >= 2: the first 5 steps lie in K-step 0)
    issue_consts(cur, 0);
    startup_skew(P, wave, lane);
    int par = 0, r0 = 0;       // constants-buffer parity, ring slot of this item's step 0

    while (true) {
        MDT_WAITV(0);
        MDT_BARRIER();
        const int xfrag = (kg * ROWS + wr * NROW + cur.a) * COLS + l31;   // halo row of output row n at tap row u: + (n + u)*COLS
        load_fw(0, r0);
        load_fx(0, xfrag, 0, 0, 0);
        const int work_n = next_valid(work + gridDim.x, nxt);
        const bool has_next = work_n < total;
        unsigned ioff_n[IS::PW];
        if (has_next) make_ioff(nxt, ioff_n);

        f32x16 acc[MW][NROW][2];   // [m][n][bb]
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NROW; ++n)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[m][n][bb][q] = 0.0f;

        // one trip = 2 K-steps = 16 steps: the register sets (fw: step parity; fx: parity of tap row + shift) and the input stage
        // are compile-time, the ring slot is rb + e (mod 6) with the trip's origin rb in a scalar register
        int rb = r0;
        for (int k2 = 0; k2 < P.NK; k2 += 2) {
            const bool last_trip = k2 + 2 >= P.NK;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kk = e >> 3, u = (e >> 2) & 1, c = e & 3, e8 = e & 7;
                const int s = (c + 1) >> 1, bb = c >> 1;
                const int k = k2 + kk;
                const int ws = e & 1, xs = (u + s) & 1;
                // ---- term 0 of this step
                MDT_PIN();
#pragma unroll
                for (int n = 0; n < NROW; ++n)
#pragma unroll
                    for (int m = 0; m < MW; ++m)
                        acc[m][n][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ws][m][1], fx[xs][n][0], acc[m][n][bb], 0, 0, 0);   // w_lo x_hi
                MDT_PIN();
                // ---- the barrier of step t publishes chunk t+1 (and, at e8 = 7, the input stage of the next K-step).  Requested by
                // this wave after chunk t+1, oldest first: {input piece, chunk (2 pieces)} of the steps t-3, t-2, t-1 -- input pieces
                // go out at e8 = 0 .. 2 from every wave (e8 = 3: waves 0 / 1 only), in front of that step's chunk:
                //     e8:  0  1  2  3  4  5  6  7
                //     N :  6  7  8  9  8  7  6  6
                const bool tail = kk == 1 && last_trip && !has_next;
                if (tail) {
                    MDT_WAITV(0);
                } else if (e8 == 0 || e8 >= 6) {
                    MDT_WAITV(6);
                } else if (e8 == 1 || e8 == 5) {
                    MDT_WAITV(7);
                } else if (e8 == 2 || e8 == 4) {
                    MDT_WAITV(8);
                } else {
                    MDT_WAITV(9);
                }
                MDT_BARRIER();
                // ---- requests of step t: one piece of the next K-step's input stage, then chunk t+5 into the slot of chunk t-1
                {
                    const bool into_next_item = kk == 1 && last_trip;      // "K-step k+1" is K-step 0 of the block's next item
                    if (e8 < IS::PW) {
                        if (!into_next_item) issue_input_piece(cur, ioff, k + 1, (kk + 1) & 1, e8);
                        else if (has_next) issue_input_piece(nxt, ioff_n, 0, 0, e8);
                    }
                    const int e5 = e8 + (R - 1), slot5 = wrap6(rb + (e + R - 1) % 6);
                    if (e5 < 8) {
                        issue_wstep(cur, k, e5 >> 2, e5 & 3, slot5);
                    } else if (!into_next_item) {
                        issue_wstep(cur, k + 1, (e5 - 8) >> 2, (e5 - 8) & 3, slot5);
                    } else if (has_next) {
                        issue_wstep(nxt, 0, (e5 - 8) >> 2, (e5 - 8) & 3, slot5);
                    }
                    if (e8 == 6 && into_next_item && has_next) issue_consts(nxt, par ^ 1);
                }
                // ---- the fragments of step t+1 (the shift s = 1 is shared by c = 1 and c = 2)
                MDT_PIN();
                if (e < 15 || !last_trip) {
                    const int e1 = (e + 1) & 15, kk1 = e1 >> 3, u1 = (e1 >> 2) & 1, c1 = e1 & 3, s1 = (c1 + 1) >> 1;
                    load_fw(ws ^ 1, wrap6(rb + (e + 1) % 6));
                    if (c1 != 2) load_fx(((u1 + s1) & 1), xfrag, kk1, u1, s1);
                }
                MDT_PIN();
                // ---- terms 1, 2 of this step
#pragma unroll
                for (int term = 1; term < 3; ++term)
#pragma unroll
                    for (int n = 0; n < NROW; ++n)
#pragma unroll
                        for (int m = 0; m < MW; ++m)
                            acc[m][n][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ws][m][0], fx[xs][n][term == 1 ? 1 : 0], acc[m][n][bb], 0, 0, 0);   // w_hi x_lo, w_hi x_hi
                MDT_PIN();
            }
            rb = wrap6(rb + 16 % 6);
        }

        EpiCtx E;
        E.res = P.res; E.y32 = P.y32; E.yrec = P.yrec;
        E.has_bias = P.bias != nullptr; E.has_act = P.yrec != nullptr && P.coef != nullptr;
        E.Cout = P.Cout; E.H = P.H; E.W = P.W; E.b = cur.b; E.kg = kg;
        E.HW = (size_t)P.H * P.W; E.planeO = (size_t)(P.H + 2) * rec_pitch(P.W); E.WpO = rec_pitch(P.W); E.dbg = pdbg(P.dbg);
        const int xi = cur.x0 + l31;
        int ys[NROW];
#pragma unroll
        for (int n = 0; n < NROW; ++n) {
            const int yi = cur.y0 + wr * NROW + n;
            ys[n] = yi < P.Hin ? 2 * yi + cur.a : P.H;      // rows past the input's last row: marked invalid
        }
        if (!(pdbg(P.dbg) & 1)) {
            epilogue_item<2, NROW, MW, 32>(E, ec_l + par * EC2, acc, wm * MW, cur.cb * MT + wm * MW, ys, 2 * xi, xi < P.Win, ResRows<NROW>{});
        }
        if (!has_next) break;
        work = work_n;
        cur = nxt;
        par ^= 1;
        r0 = rb;                                             // the ring runs on: slot of the next item's step 0
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) ioff[i] = ioff_n[i];
    }
}

}  // namespace

namespace mdt {

// arrival counters of startup_skew: one buffer per device, allocated on first use (under a lock: launches may come from several host
// threads), zeroed once; every launch takes a fresh epoch (see startup_skew), so nothing is ever reset.  The counter words are per
// DEVICE, not per stream: two rec2 launches in flight on one device at once (different streams) re-stamp each other's arrival counts,
// so "second block on this CU" can be decided wrongly for either -- that moves a block's START by half an item, never a result (the
// plugin and the bench run one stream per device; the heuristic assumes that).
static unsigned* cu_counters(unsigned* epoch) {
    static std::mutex mu;
    static unsigned* buf[64] = {};
    static unsigned next_epoch[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!buf[dev]) {
        unsigned* p = nullptr;
        if (hipMalloc(&p, 16 * 256 * sizeof(unsigned)) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 16 * 256 * sizeof(unsigned)) != hipSuccess) {
            (void)hipFree(p);      // (no counters: this launch runs without the skew; the next one tries again)
            return nullptr;
        }
        buf[dev] = p;
    }
    next_epoch[dev] = (next_epoch[dev] + 1u) & 0xFFFFFFu;
    if (next_epoch[dev] == 0u) next_epoch[dev] = 1u;      // (0 = the zeroed buffer's epoch)
    *epoch = next_epoch[dev];
    return buf[dev];
}

// Probing switches (PROBES build of the library only -- common.h: probe_env; read per launch so that a probe can flip them in-process):
//   MDTILE_REC2_SKEW    0 = no start-up delay, 1 = by block index (>= grid / 2), 2 = by the per-CU arrival counter (default)
//   MDTILE_REC2_SKEW_PCT  the delay as a percentage of an item's K loop at one block per CU-half (default 100)
//   MDTILE_REC2_CENSUS  device address (hex) of a [grid] unsigned buffer that receives every block's hardware CU id
int conv_rec2_launch(ConvRParams P, int B, int up, hipStream_t s, int cus) {
    int skew = 2, pct = 100;
    if (const char* e = probe_env("MDTILE_REC2_SKEW")) skew = atoi(e);
    if (const char* e = probe_env("MDTILE_REC2_SKEW_PCT")) pct = atoi(e);
    P.census = nullptr;
    if (const char* e = probe_env("MDTILE_REC2_CENSUS")) P.census = reinterpret_cast<unsigned*>((uintptr_t)strtoull(e, nullptr, 16));
    P.epoch = 0;
    P.cu_ctr = skew == 2 ? cu_counters(&P.epoch) : nullptr;
    int per_cu = 2;                                   // two blocks per CU
    if (const char* e = probe_env("MDTILE_REC2_PER_CU")) per_cu = atoi(e) == 1 ? 1 : 2;      // probing: a 4-wave block alone on its CU
    const int grid_max = per_cu * (cus / 8 * 8);
    if (up) {
        // K loop of one item with the SIMDs to itself: NK x 8 steps x 12 MFMAs x 32 clk at ~2 GHz = NK x 1.5 us; in 10 ns ticks
        P.skew_ticks = skew ? (unsigned)((long long)P.NK * 154 * pct / 100) : 0u;
        P.PX = (P.Win + 31) / 32;
        P.ptiles = P.PX * ((P.Hin + 3) / 4);
        const long long items = (long long)((P.ptiles + 7) / 8) * 8 * P.NCB * 2 * B;
        dim3 grid((unsigned)(items < grid_max ? items : grid_max)), block(256);
        hipLaunchKernelGGL(k_upconv_rec2, grid, block, 0, s, P);
        MDT_LAUNCH_CHECK();
        return MDTILE_OK;
    }
    // NK x 9 steps x 24 MFMAs x 32 clk = NK x 3.5 us
    P.skew_ticks = skew ? (unsigned)((long long)P.NK * 346 * pct / 100) : 0u;
    P.PX = (P.W + 31) / 32;
    P.ptiles = P.PX * ((P.H + 7) / 8);
    const long long items = (long long)((P.ptiles + 7) / 8) * 8 * P.NCB * B;
    dim3 grid((unsigned)(items < grid_max ? items : grid_max)), block(256);
    hipLaunchKernelGGL((k_conv3x3_rec2<2, 2, 4>), grid, block, 0, s, P);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

}  // namespace mdt
