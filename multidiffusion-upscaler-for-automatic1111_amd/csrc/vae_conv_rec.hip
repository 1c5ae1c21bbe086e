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

#include "conv_rec_common.h"

namespace {

// One unit of work of a persistent block: a (batch sample, pixel tile, cout block) triple.
struct WorkItem {
    int b, cb, y0, x0;
};

// direct 3x3:  MW = 32-cout tiles per wave (2), WM = waves along cout, NROW = pixel rows per wave (4: two half-steps of 2).
// PERSISTENT: the grid is one block per CU; block i works through items i, i + grid, i + 2 grid, ...  The first operands of
// the NEXT item (weight chunks 0 and 1, input tile of K-step 0) are DMA'd behind the LAST barrier of the current item's K loop
// -- their LDS slots are free by then -- so neither the launch gap nor the prologue's HBM round trip shows between items:
// they run under the last MFMAs and the epilogue.
// Start-up stagger of the persistent blocks (ConvRParams::skew_ticks = the spread in 100 MHz ticks; 0 = off).  Every block runs items of
// ONE duration, so the whole chip moves in lock step: all 256 CUs spend the K loop without touching HBM and then all write (and, for a
// conv2, read) their epilogue at once -- 64 MB per tensor and round, served at the HBM rate while the matrix cores idle (round 4:
// probes/conv_item_timeline.py).  A block that starts `d` late stays `d` late for the whole launch: with the
// delays spread over an item period the epilogue bursts of the CUs fall under the K loops of the others.  Price: the last block ends a
// spread later -- the launcher staggers only launches of many rounds.  Wave 0 sleeps; the other waves wait at the first barrier.
__device__ __forceinline__ void stagger_start(const ConvRParams& P, int wave) {
    if (wave != 0 || P.skew_ticks == 0) return;
    const unsigned frac = (blockIdx.x * 0x9E3779B1u) >> 24;                    // 0 .. 255, scattered over the block indices (XCDs, CUs)
    const unsigned long long d = ((unsigned long long)P.skew_ticks * frac) >> 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < d) __builtin_amdgcn_s_sleep(32);
}

template <int MW, int WM, int NROW>
__global__ __launch_bounds__(512, 2) void k_conv3x3_rec(const ConvRParams P) {
    constexpr int WR = 8 / WM, TH = WR * NROW, MT = MW * WM, HN = NROW / 2;
    constexpr int ROWS = TH + 2, COLS = 34;
    using IS = InStage<ROWS>;
    constexpr int W_REC = 2 * 3 * MT * 64;              // [hl][dx][mt][lane]
    constexpr int W_DMA = W_REC / 64;
    constexpr int W_PW = (W_DMA + 7) / 8;
    __shared__ u32x4 smem[2 * IS::PAD + 3 * W_REC + 2 * EC_REC];
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + 2 * IS::PAD;
    u32x4* const ec_l = smem + 2 * IS::PAD + 3 * W_REC;     // per-channel epilogue constants, two buffers (item parity)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wr = wave / WM;
    const int Hp = P.H + 2, Wp = rec_pitch(P.W), Pn = P.Cin >> 3;
    const size_t plane = (size_t)Hp * Wp;

    // work -> (sample, pixel tile, cout block).  Workgroups go to XCDs round-robin (id % 8) and grid % 8 == 0 whenever a block
    // sees more than one item, so `work % 8` is this block's XCD for all its items: all cout blocks of a pixel tile stay on one L2.
    const int per_img = ((P.ptiles + 7) / 8) * 8 * P.NCB, total = per_img * P.B;
    auto decode = [&](int work, WorkItem& it) -> bool {
        it.b = work / per_img;
        const int r = work - it.b * per_img, xcd = r & 7, slot = r >> 3;
        const int ptile = (slot / P.NCB) * 8 + xcd;
        it.cb = slot % P.NCB;
        const int py = ptile / P.PX, px = ptile - py * P.PX;
        it.y0 = py * TH;
        it.x0 = px * 32;
        return ptile < P.ptiles;
    };
    auto next_valid = [&](int work, WorkItem& it) -> int {   // first item >= work (stride grid) that is a real tile, or >= total
        while (work < total && !decode(work, it)) work += gridDim.x;
        return work;
    };

    // input DMA map: wave-instruction di = wave + 8 i covers LDS records [64 di, 64 di + 64) of a stage; hl = di / HALF_DMA
    auto make_ioff = [&](const WorkItem& it, unsigned (&ioff)[IS::PW]) {   // byte offsets inside the (K-step, hl) pair of planes
        int ln = lane;
        asm volatile("" : "+v"(ln));      // the (g, r, c) of a piece are re-derived per item (~10 VALU each): kept across the persistent loop they are
                                          // 15 registers that the epilogue's peak (accumulators + two residual buffers) pushes into scratch
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            const int di = wave + 8 * i;
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
    auto issue_input = [&](const WorkItem& it, const unsigned (&ioff)[IS::PW], int k, int stage) {
        const char* xb = reinterpret_cast<const char*>(P.x + (size_t)it.b * 2 * Pn * plane);
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            const int di = wave + 8 * i;
            if (di < IS::DMA) {
                const char* base = xb + ((size_t)(di / IS::HALF_DMA) * Pn + 2 * (size_t)k) * plane * 16;   // wave-uniform
                dma16(base, ioff[i], in_l + stage * IS::PAD + di * 64);
            }
        }
    };
    const unsigned lane16 = lane * 16;
    auto issue_weights = [&](const WorkItem& it, int ph, int ring) {
        const char* wsrc = reinterpret_cast<const char*>(P.w + (size_t)it.cb * P.NK * 3 * W_REC);
#pragma unroll
        for (int i = 0; i < W_PW; ++i)
            if (wave + 8 * i < W_DMA) {
                const char* base = wsrc + ((size_t)ph * W_REC + (wave + 8 * i) * 64) * 16;
                dma16(base, lane16, w_l + ring * W_REC + (wave + 8 * i) * 64);
            }
    };

    // epilogue constants of an item's BM couts: waves 0 / 1 / 2 fetch bias / a / s (512 B each; the upper lanes repeat the
    // lower ones into the pad half of the 1 KB slot)
    const unsigned lane16h = (lane % (MT * 8)) * 16;      // MT * 32 floats = MT * 8 lanes x 16 B; the other lanes repeat them
    auto issue_consts = [&](const WorkItem& it, int par) {
        if (wave == 0 && P.bias) dma16(reinterpret_cast<const char*>(P.bias + it.cb * (MT * 32)), lane16h, ec_l + par * EC_REC);
        if ((wave == 1 || wave == 2) && P.yrec && P.coef)
            dma16(reinterpret_cast<const char*>(P.coef + ((size_t)it.b * 2 + (wave - 1)) * P.Cout + it.cb * (MT * 32)), lane16h,
                  ec_l + par * EC_REC + wave * 64);
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

    WorkItem cur, nxt;
    int work = next_valid(blockIdx.x, cur);
    if (work >= total) return;
    unsigned ioff[IS::PW];
    make_ioff(cur, ioff);
    issue_input(cur, ioff, 0, 0);
    issue_weights(cur, 0, 0);
    issue_weights(cur, 1, 1);
    issue_consts(cur, 0);
    stagger_start(P, wave);
    const int nph = P.NK * 3;
    int par = 0;

    // probing (MDTILE_REC_DBG bit 3 + MDTILE_REC_STAMPS=<device address>): block 0 records s_memtime per wave and item at
    //   0 item start (behind the barrier) | 1 K loop done | 2 epilogue code done (stores issued) | 4 vmcnt(0) + barrier of the next item passed
    unsigned long long* const stamps = (pdbg(P.dbg) & 8) && P.census && blockIdx.x == 0 ? reinterpret_cast<unsigned long long*>(P.census) : nullptr;
    int item_no = 0;
    auto stamp = [&](int k) {
        if (stamps && lane == 0 && item_no < 64) stamps[(item_no * 8 + wave) * 8 + k] = __builtin_readcyclecounter();
    };
    // a conv2's residual arrives in the accumulators (conv_rec_common.h: ResRows): the first item's rows are requested here, every later
    // item's by the epilogue of the item before it
    f32x16 acc[MW][NROW][1];
    const bool res_in_acc = P.res != nullptr && !(pdbg(P.dbg) & 1);
    auto res_rows = [&](const WorkItem& it, bool on) {
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
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of the item's first operands have landed (and the residual rows)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (item_no > 0) { --item_no; stamp(4); ++item_no; }
        stamp(0);
        load_fw(0, 0, 0);
        load_fx(0, 0, 0, 0, 0);
        const int work_n = next_valid(work + gridDim.x, nxt);
        unsigned ioff_n[IS::PW];
        if (work_n < total) make_ioff(nxt, ioff_n);      // (outside the unrolled K loop: keeps its body under the unroll budget)

        if (!res_in_acc) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NROW; ++n)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[m][n][0][q] = 0.0f;
        }

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
                            acc[m][h * HN + n][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ws][m][term == 0 ? 1 : 0], fx[xs][n][term == 1 ? 1 : 0],
                                                                                            acc[m][h * HN + n][0], 0, 0, 0);   // w_lo x_hi, w_hi x_lo, w_hi x_hi
                MDT_PIN();
                if (dx == 0 && h == 1) {
                    // Pieces this wave has in flight, oldest first: weight chunk ph+1 (issued behind the previous barrier) and, at
                    // dy = 1, the input tile of K-step k+1 issued right after it.  The weights are read from the end of this phase
                    // on, the input tile only from the end of the dy = 2 phase: at dy = 1 the IS::PW youngest pieces (every wave
                    // issues exactly that many, IS::DMA % 8 == 0) may stay in flight -- a whole extra phase for their HBM round trip.
                    static_assert(IS::DMA % 8 == 0 && IS::PW == 5, "the counted wait below assumes 5 input pieces per wave");
                    if (dy == 1 && k + 1 < P.NK) __builtin_amdgcn_s_waitcnt(0x0F75);   // vmcnt(5)
                    else __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0)
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                // DMA issue of this phase (~10 scalar / VMEM instructions per piece), staggered between the two waves that share a
                // SIMD (w and w + 4): waves 0-3 right behind the barrier, waves 4-7 one half-step later -- while one of the pair
                // issues its pieces the other one keeps the matrix pipe fed.
                // Ring slot of chunk ph+2 = the one chunk ph-1 held: every wave finished reading it before the barrier.
                if ((dx == 0 && h == 1 && wave < 4) || (dx == 1 && h == 0 && wave >= 4)) {
                    if (ph + 2 < nph) issue_weights(cur, ph + 2, (dy + 2) % 3);
                    if (dy == 0 && k + 1 < P.NK) issue_input(cur, ioff, k + 1, (kk + 1) & 1);
                    if (kk == 1 && dy == 2 && k + 1 == P.NK && work_n < total) {
                        // last phase of the item: ring slots 0 / 1 and input stage 0 are out of use (NK is even) -> the next
                        // item's first operands go there now and land under the remaining MFMAs and the epilogue
                        issue_input(nxt, ioff_n, 0, 0);
                        issue_weights(nxt, 0, 0);
                        issue_weights(nxt, 1, 1);
                        issue_consts(nxt, par ^ 1);
                    }
                }
            }
        }

        stamp(1);
        EpiCtx E;
        E.res = P.res; E.y32 = P.y32; E.yrec = P.yrec;
        E.has_bias = P.bias != nullptr; E.has_act = P.yrec != nullptr && P.coef != nullptr;
        E.Cout = P.Cout; E.H = P.H; E.W = P.W; E.b = cur.b; E.kg = kg;
        E.HW = (size_t)P.H * P.W; E.planeO = plane; E.WpO = Wp;
        E.dbg = pdbg(P.dbg);
        int le = lane;
        asm volatile("" : "+v"(le));      // (re-derived: a separate l31 kept alive through the epilogue goes to scratch)
        const int x = cur.x0 + (le & 31);
        int ys[NROW];
#pragma unroll
        for (int n = 0; n < NROW; ++n) ys[n] = cur.y0 + wr * NROW + n;
        if (!(pdbg(P.dbg) & 1)) {
            epilogue_item<1, NROW, MW>(E, ec_l + par * EC_REC, acc, wm * MW, cur.cb * MT + wm * MW, ys, x, x < P.W, res_rows(nxt, work_n < total));
        }
        stamp(2);
        ++item_no;
        if (work_n >= total) break;
        work = work_n;
        cur = nxt;
        par ^= 1;
        for (int i = 0; i < IS::PW; ++i) ioff[i] = ioff_n[i];
    }
}

// =====================================================================================================================
// nearest-2x upsample + 3x3 conv in sub-pixel form (four 2x2 convs on the un-upsampled grid, see vae_conv_bf16x3.hip
// k_upconv_bf16x3 for the derivation).  Item = 128 couts x (8 x 32 INPUT px) of ONE output-row parity a and both column
// parities bb; phases (K-step k, tap row u); a phase = 4 combo-steps c of 12 MFMAs per wave:
//     c:  0 (shift s 0, bb 0)   1 (s 1, bb 0)   2 (s 1, bb 1)   3 (s 2, bb 1)        tap column v = s - bb
// Weight chunk [hl][bb][v][mt][lane] per (a, cb, k, u) in a 3-slot ring.  Persistent like k_conv3x3_rec; an item has 2 NK
// phases, which need not be a multiple of 3, so the ring position of an item's first chunk (r0) rotates from item to item:
// the next item's chunks 0 / 1 go to the two slots the last phase is NOT reading.
__global__ __launch_bounds__(512, 2) void k_upconv_rec(const ConvRParams P) {
    constexpr int MT = 4, MW = 2, WM = 2, NROW = 2, TH = 8;
    constexpr int ROWS = TH + 2, COLS = 34;
    using IS = InStage<ROWS>;
    constexpr int W_REC = 2 * 2 * 2 * MT * 64, W_DMA = W_REC / 64, W_PW = W_DMA / 8;
    __shared__ u32x4 smem[2 * IS::PAD + 3 * W_REC + 2 * EC_REC];
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + 2 * IS::PAD;
    u32x4* const ec_l = smem + 2 * IS::PAD + 3 * W_REC;

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
            const int di = wave + 8 * i;
            int s = (di % IS::HALF_DMA) * 64 + lane;
            if (s >= IS::HALF) s = IS::HALF - 1;
            const int g = s / (ROWS * COLS), p = s - g * (ROWS * COLS);
            const int r = p / COLS, c = p - r * COLS;
            int pr = P.iy0[it.b & (REC_WIN_MAXB - 1)] + it.y0 + r, pc = P.ix0[it.b & (REC_WIN_MAXB - 1)] + it.x0 + c;     // inside the window's own border: the image's real neighbours
            pr = pr < Hp ? pr : Hp - 1;
            pc = (pc < P.WinF + 1 ? pc : P.WinF + 1) + REC_COL0;
            ioff[i] = (unsigned)(((size_t)g * plane + (size_t)pr * Wp + pc) * 16);
        }
    };
    auto issue_input = [&](const Item& it, const unsigned (&ioff)[IS::PW], int k, int stage) {
        const char* xb = reinterpret_cast<const char*>(P.x + (size_t)it.b * 2 * Pn * plane);
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
    const unsigned lane16 = lane * 16;
    auto issue_weights = [&](const Item& it, int ph, int ring) {
        const char* wsrc = reinterpret_cast<const char*>(P.w + ((size_t)it.a * P.NCB + it.cb) * nph * W_REC);
#pragma unroll
        for (int i = 0; i < W_PW; ++i) {
            const char* base = wsrc + ((size_t)ph * W_REC + (wave + 8 * i) * 64) * 16;
            dma16(base, lane16, w_l + ring * W_REC + (wave + 8 * i) * 64);
        }
    };
    const unsigned lane16h = (lane & 31) * 16;
    auto issue_consts = [&](const Item& it, int par) {
        if (wave == 0 && P.bias) dma16(reinterpret_cast<const char*>(P.bias + it.cb * (MT * 32)), lane16h, ec_l + par * EC_REC);
        if ((wave == 1 || wave == 2) && P.yrec && P.coef)
            dma16(reinterpret_cast<const char*>(P.coef + ((size_t)it.b * 2 + (wave - 1)) * P.Cout + it.cb * (MT * 32)), lane16h,
                  ec_l + par * EC_REC + wave * 64);
    };

    bf16x8 fw[2][MW][2];     // [set][m][hl]   weight tiles of one combo-step
    bf16x8 fx[2][NROW][2];   // [set][n][hl]   input rows of one column shift
    const int wfrag = wm * MW * 64 + lane;
    auto load_fw = [&](int set, int ring, int c) {
        const int bb = c >> 1, v = ((c + 1) >> 1) - bb;           // c: 0 -> (0, 0), 1 -> (0, 1), 2 -> (1, 0), 3 -> (1, 1)
        const u32x4* wst = w_l + ring * W_REC + wfrag;
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fw[set][m][hl] = __builtin_bit_cast(bf16x8, wst[(((hl * 2 + bb) * 2 + v) * MT + m) * 64]);
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
    issue_input(cur, ioff, 0, 0);
    issue_weights(cur, 0, 0);
    issue_weights(cur, 1, 1);
    issue_consts(cur, 0);
    stagger_start(P, wave);
    int par = 0, r0 = 0;       // constants-buffer parity, ring slot of this item's chunk 0

    while (true) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int xfrag = (kg * ROWS + wr * NROW + cur.a) * COLS + l31;   // halo row of output row n at tap row u: + (n + u)*COLS
        int rs[3];                                                        // ring slot of local phase p: rs[p % 3]
#pragma unroll
        for (int i = 0; i < 3; ++i) rs[i] = (r0 + i) % 3;
        load_fw(0, rs[0], 0);
        load_fx(0, xfrag, 0, 0, 0);
        const int work_n = next_valid(work + gridDim.x, nxt);
        unsigned ioff_n[IS::PW];
        if (work_n < total) make_ioff(nxt, ioff_n);
        const int r0_n = (r0 + nph) % 3;                                  // = (slot of the last chunk + 1) % 3

        f32x16 acc[MW][NROW][2];   // [m][n][bb]
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NROW; ++n)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[m][n][bb][q] = 0.0f;

        // one trip = 3 K-steps = 6 phases = 24 combo-steps: the register sets are compile-time (fw set = combo-step parity; fx set =
        // parity of the running shift counter 3*phase + s) and so is the ring index modulo the item's r0.  NK % 3 != 0: the surplus
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
                        load_fw(ws ^ 1, rs[pl_ % 3], c + 1);
                        if (c != 1) load_fx(xs ^ 1, xfrag, k & 1, u, s + 1);
                    } else {
                        const int pl1 = (pl_ + 1) % 6, kk1 = pl1 >> 1, u1 = pl1 & 1;
                        const int k1 = (pl_ < 5 ? k3 : k3 + 3) + kk1;
                        if (k1 < P.NK) {
                            load_fw(ws ^ 1, rs[pl1 % 3], 0);
                            load_fx(xs ^ 1, xfrag, k1 & 1, u1, 0);
                        }
                    }
                    MDT_PIN();
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int n = 0; n < NROW; ++n)
#pragma unroll
                            for (int m = 0; m < MW; ++m)
                                acc[m][n][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ws][m][term == 0 ? 1 : 0], fx[xs][n][term == 1 ? 1 : 0],
                                                                                        acc[m][n][bb], 0, 0, 0);
                    MDT_PIN();
                    if (c == 1) {
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    // staggered DMA issue (see k_conv3x3_rec): waves 0-3 behind the barrier, waves 4-7 one combo-step later
                    if ((c == 1 && wave < 4) || (c == 2 && wave >= 4)) {
                        if (ph + 2 < nph) issue_weights(cur, ph + 2, rs[(pl_ + 2) % 3]);
                        if (u == 0 && k + 1 < P.NK) issue_input(cur, ioff, k + 1, (k + 1) & 1);
                        if (ph + 1 == nph && work_n < total) {
                            // last phase: the two ring slots it does not read and input stage 0 (NK is even: the last K-step sits in
                            // stage 1) take the next item's first operands
                            issue_input(nxt, ioff_n, 0, 0);
                            issue_weights(nxt, 0, r0_n);
                            issue_weights(nxt, 1, (r0_n + 1) % 3);
                            issue_consts(nxt, par ^ 1);
                        }
                    }
                }
            }
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
            epilogue_item<2, NROW, MW>(E, ec_l + par * EC_REC, acc, wm * MW, cur.cb * MT + wm * MW, ys, 2 * xi, xi < P.Win, ResRows<NROW>{});
        }
        if (work_n >= total) break;
        work = work_n;
        cur = nxt;
        par ^= 1;
        r0 = r0_n;
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) ioff[i] = ioff_n[i];
    }
}

// =====================================================================================================================
// fp32 NCHW -> record image (+ optional fixed-statistics GroupNorm + SiLU): entry points of the record path (conv_in /
// attention outputs, the fast-mode estimator and slow mode, where the statistics only exist after the producer ran).
// One thread = one (pixel, plane) of the PADDED image; border threads write the zero records.
__global__ __launch_bounds__(256) void k_rec_from_f32(const float* __restrict__ x, const float* __restrict__ coef, u32x4* __restrict__ rec,
                                                      int C, int H, int W) {
    const int Wp = rec_pitch(W), Hp = H + 2, Pn = C >> 3;
    const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y;      // padded coordinates: (0, 0) = the top-left border record
    const int bp = blockIdx.z, b = bp / Pn, p = bp - b * Pn;
    if (px >= W + 2) return;
    const size_t planeO = (size_t)Hp * Wp;
    u32x4* hi_p = rec + ((size_t)b * 2 * Pn + p) * planeO + (size_t)py * Wp + px + REC_COL0;
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
    const int Wp = rec_pitch(W), Hp = H + 2, Pn = C >> 3;
    const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y;
    const int bp = blockIdx.z, b = bp / Pn, p = bp - b * Pn;
    if (px >= W) return;
    const size_t planeO = (size_t)Hp * Wp;
    const u32x4* hi_p = rec + ((size_t)b * 2 * Pn + p) * planeO + (size_t)(py + 1) * Wp + (px + 1 + REC_COL0);
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

// MDTILE_REC_PERSIST=0 (probes build only): one item per block (A/B of the persistent schedule; read per launch so a probe can flip it in-process)
constexpr int REC_STAGGER_PCT_DEFAULT = 0;      // off: -2 ... -5 % on single conv1 launches (profiles/r4u, r4v), nothing on the 8K decode (profiles/r4y: 1869 / 1870 vs 1870 / 1874 ms)

static bool rec_persistent() {
    const char* e = probe_env("MDTILE_REC_PERSIST");
    return !(e && e[0] == '0');
}

// MDTILE_REC_GRID=n: probing -- the persistent kernels run as if the chip had n CUs (n % 8 == 0)
static int num_cus() {
    if (const char* e = probe_env("MDTILE_REC_GRID")) {
        const int n = atoi(e);
        if (n >= 8) return n / 8 * 8;
    }
    static const int n = [] {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        return cus;
    }();
    return n;
}

// vae_conv_rec2.hip: the two-blocks-per-CU form of the cout % 128 == 0 kernels
int conv_rec2_launch(ConvRParams P, int B, int up, hipStream_t s, int cus);
// vae_conv_recd.hip: 64-cout items with the epilogue dripped into the next item's K loop (direct 3x3, cin % 128 == 0)
bool conv_recd_supported(int cout, int cin);
int conv_recd_launch(ConvRParams P, int B, hipStream_t s, int cus);

// Which kernel family takes a launch.  The two-blocks-per-CU kernels (vae_conv_rec2.hip) lose 3-12 % on launches that fill the chip many
// times over (profiles/r4a: their 8-row items double the weight stream through the CU's memory pipe and the store epilogue is not
// hidden), but their items are half as large and 512 of them are resident: launches of only a few item rounds quantise better
// (512 -> 512 at 86 x 86 x 3 tiles: +56 %, at 278 x 278: +2 ... +12 %; at 256 x 256, an exact fit of the one-block grid: -5 %).
// Cost model, in units of one 16-row item on a CU of its own (fitted on profiles/r4d/conv_two_blocks_small_launches.log):
//   one block / CU:   ceil(items16 / CUs)
//   two blocks / CU:  full rounds of 2 CUs items cost `pair`; a last round of <= CUs items (each block alone on its CU) costs `lone`
// family = 1 / 2 (MDTILE_CONV_REC_ONE_BLOCK / _TWO_BLOCKS in the call's flags) names the family (tests, A/B in probes/conv_rec2_ab.py).
static bool rec_two_blocks(long long items16, long long items8, int cus, int up, int family) {
    if (family == 1) return false;
    if (family == 2) return true;
    if (const char* e = probe_env("MDTILE_REC_BLOCKS")) {      // (probes build only: the old in-process A/B switch of the scripts under probes/)
        if (e[0] == '1') return false;
        if (e[0] == '2') return true;
    }
    const double pair = up ? 1.13 : 1.05, lone = up ? 0.62 : 0.60;
    const double t1 = (double)((items16 + cus - 1) / cus);
    const long long full = items8 / (2 * cus), rem = items8 - full * 2 * cus;
    const double t2 = full * pair + (rem == 0 ? 0.0 : rem <= cus ? lone : pair);
    return t2 < 0.97 * t1;
}

bool conv_rec_supported(int cout, int cin, int ksize) { return ksize == 3 && cin % 32 == 0 && (cout % 128 == 0 || (cout >= 1 && cout < 32)); }

size_t rec_image_bytes(int B, int C, int H, int W) { return (size_t)B * C * (H + 2) * rec_pitch(W) * 4; }

size_t rec_plane_records(int H, int W) { return (size_t)(H + 2) * rec_pitch(W); }

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

// win (sub-pixel upsample kernel only, else null): {HinF, WinF, y0[0], x0[0], ..., y0[7], x0[7]} -- d_xrec is the record image of
// [B, cin, HinF, WinF] and image b's conv reads its window [y0[b & 7] : .. + H/2, x0[b & 7] : .. + W/2]  (all 8 slots filled)
int conv_rec_launch(const void* d_xrec, const void* d_w_rec, const float* d_bias, const float* d_res, float* d_y32, void* d_yrec,
                    const float* d_ycoef, int B, int cin, int cout, int H, int W, int up, hipStream_t s, const int* win, int family) {
    ConvRParams P;
    P.x = (const u32x4*)d_xrec; P.w = (const u32x4*)d_w_rec; P.bias = d_bias; P.res = d_res; P.y32 = d_y32;
    P.yrec = (u32x4*)d_yrec; P.coef = d_ycoef;
    P.B = B; P.Cin = cin; P.Cout = cout; P.H = H; P.W = W;
    P.Hin = up ? H / 2 : H; P.Win = up ? W / 2 : W;
    P.HinF = win ? win[0] : P.Hin; P.WinF = win ? win[1] : P.Win;
    for (int b = 0; b < REC_WIN_MAXB; ++b) {       // (image b reads slot b & 7; the caller fills all 8)
        P.iy0[b] = win ? win[2 + 2 * b] : 0;
        P.ix0[b] = win ? win[3 + 2 * b] : 0;
    }
    P.NCB = cout % 128 == 0 ? cout / 128 : 1;
    P.NK = cin / 16;
    P.skew_ticks = 0; P.cu_ctr = nullptr; P.census = nullptr; P.epoch = 0;
    P.dbg = 0;
    if (const char* e = probe_env("MDTILE_REC_DBG")) P.dbg = atoi(e);      // probing only (probes/conv_rec_diag.py, conv_item_timeline.py): see ConvRParams::dbg
    if (P.dbg & 8)
        if (const char* e = probe_env("MDTILE_REC_STAMPS")) P.census = reinterpret_cast<unsigned*>((uintptr_t)strtoull(e, nullptr, 16));
    if (up) P.w = (const u32x4*)d_w_rec + conv_bf16x3_direct_records(cout, cin);
    if (family == 3 && !up && conv_recd_supported(cout, cin)) return conv_recd_launch(P, B, s, num_cus());
    if (cout % 128 == 0 && rec_persistent()) {
        const int cus = num_cus() / 8 * 8;
        const int hin = up ? P.Hin : H, win = up ? P.Win : W, per = P.NCB * (up ? 2 : 1);      // items tile the INPUT grid of the sub-pixel form
        const long long px = (win + 31) / 32;
        const long long items16 = (px * ((hin + (up ? 7 : 15)) / (up ? 8 : 16)) + 7) / 8 * 8 * per * B;
        const long long items8 = (px * ((hin + (up ? 3 : 7)) / (up ? 4 : 8)) + 7) / 8 * 8 * per * B;
        if (rec_two_blocks(items16, items8, cus, up, family)) return conv_rec2_launch(P, B, up, s, num_cus());
    }
    // start-up stagger (stagger_start): spread = MDTILE_REC_STAGGER_PCT percent of an estimated item period, launches of >= 6 rounds only
    // default: a quarter period for launches that write records ONLY (a conv1: -2 ... -5 % per launch, profiles/r4u, r4v; launches with an fp32
    // stream run at their CU's own memory-pipe floor either way and only pay the late end)
    auto stagger = [&](long long items, int cus, unsigned period_ticks) {
        const char* e = probe_env("MDTILE_REC_STAGGER_PCT");      // (read per launch: probes/conv_stagger_ab.py switches it between launches)
        const int p = e ? atoi(e) : ((!d_y32 && !d_res) ? REC_STAGGER_PCT_DEFAULT : 0);
        P.skew_ticks = (rec_persistent() && items >= 6LL * cus && p > 0) ? period_ticks * (unsigned)p / 100u : 0u;
    };
    if (up) {
        P.PX = (P.Win + 31) / 32;
        P.ptiles = P.PX * ((P.Hin + 7) / 8);
        const long long items = (long long)((P.ptiles + 7) / 8) * 8 * P.NCB * 2 * B;
        const int cus = num_cus();
        stagger(items, cus, (unsigned)P.NK * 400u + 2000u);
        dim3 grid((unsigned)((items < cus || !rec_persistent()) ? items : cus / 8 * 8)), block(512);
        hipLaunchKernelGGL(k_upconv_rec, grid, block, 0, s, P);
        MDT_LAUNCH_CHECK();
        return MDTILE_OK;
    }
    P.PX = (W + 31) / 32;
    P.ptiles = P.PX * ((H + 15) / 16);
    const long long items = (long long)((P.ptiles + 7) / 8) * 8 * P.NCB * B;
    const int cus = num_cus();                    // one block per CU (155 KB LDS, 2 waves per SIMD)
    if (cout % 128 == 0) stagger(items, cus, (unsigned)P.NK * 900u + 1500u);
    dim3 grid((unsigned)((items < cus || !rec_persistent()) ? items : cus / 8 * 8)), block(512);
    if (cout % 128 == 0) hipLaunchKernelGGL((k_conv3x3_rec<2, 2, 4>), grid, block, 0, s, P);
    else hipLaunchKernelGGL((k_conv3x3_rec<1, 1, 2>), grid, block, 0, s, P);      // conv_out: one 32-cout tile, bias padded to 32 by the caller
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

}  // namespace mdt
