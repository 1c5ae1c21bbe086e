// Record-image 3x3 conv with a DRIPPED (software-pipelined) epilogue  (round 5).
//
// Same arithmetic, record images, packed weights and per-accumulator MFMA order as vae_conv_rec.hip (results bit-identical to
// k_conv3x3_rec<2,2,4>); what changes is WHEN an item's results leave the CU.  There an item's 8 waves finish their K loop
// together and then spend 14-46 k cycles (record / fp32 + records + residual) issuing bias, activation, split and stores with the
// matrix cores idle: the CU's vector-memory pipe is ONE in-order queue that retires ~27 B/clk of stores (DESIGN.md section 3,
// "Late round 4"), a conv2 item moves 768 KB through it, and since gfx9 counts stores in vmcnt the next item's operand DMA waits
// sit behind them.  Two resident blocks did not hide it either (vae_conv_rec2.hip: the other block's DMA queues behind the same
// stores).  What was never tried is to keep the stores from piling up at all:
//
//   * a wave holds TWO accumulator sets: `acc` (the item whose K loop is running) and `sealed` (the results of the block's
//     PREVIOUS item).  Item i-1's bias / fp32 stores / activation / split / record stores and the residual loads of item i+1 are
//     issued in SLOTS of <= 4 KB per wave between the K-steps of item i -- at most one slot per phase, never in a phase that also
//     requests an input stage -- so the memory pipe sees ~7 B/clk of epilogue traffic next to ~11 B/clk of operand DMA and never
//     queues; the slot's VALU work (2 transcendentals per value) runs in the issue slack a wave has beside the MFMAs of the wave
//     it shares its SIMD with (waves w and w + 4 take their slots one step apart).
//   * both sets have to fit the 256-register budget of two waves per SIMD next to two fragment sets: 64 + 64 accumulator
//     registers = four 32x32 tiles each.  The item is therefore 64 couts x 16 rows x 32 px (wave tile 64 couts x 2 rows), HALF
//     the couts of k_conv3x3_rec's item on the same 18-row input stage: the input stream from L2 doubles per MFMA, the weight
//     stream halves (9.5 vs 7.1 KB of DMA per K-step and 16 rows: 1.35x -- the row-halved items of vae_conv_rec2.hip pay 1.69x).
//   * at the item boundary the two sets change roles (64 register swaps per wave); the residual of a conv2 is loaded into the
//     sealed registers right after they were stored from, so the next K loop starts from it: y = (res + sum) + bias as before.
//   * the block's LAST item has no K loop to hide under: its epilogue runs in one piece (1 item in ~70).
//
// K loop: the protocol of k_conv3x3_rec -- phases (K-step k, tap row dy) of three steps dx (12 MFMAs per wave and step), input
// stage [hl][kg][18][34] records x 2, weight chunks [hl][dx][mt 2][lane] (12 KB) in a 3-slot ring (slot = dy), ONE barrier per
// phase, chunk ph + 2 and the next K-step's input requested behind it, the next item's first operands in the item's last phase.
// An item's K loop = the SLOT TRIP (K-steps 0-3, 36 steps unrolled) + plain trips of two K-steps (18 steps unrolled; cin % 32 == 0,
// cin >= 128).  The 8 slots of an item sit in the dy = 1 / dy = 2 phases of the slot trip, position i = 2 k + dy - 1, for the
// wave's four (m-tile, row) units u = i / 2, each slot in THREE pieces behind the phase's three MFMA blocks:
//     i even: + bias, fp32 stores | activation of channels 0-7 of the lane's 16 | split + record stores of them
//     i odd : activation of channels 8-15 | split + record stores (+ the item's zero border cells) | residual row of item i+1 -> registers
// (a dy = 0 phase spreads the 5 input pieces of the next K-step over its three pieces instead).  No piece is longer than the
// 12-MFMA block the wave's SIMD partner covers it with; the two waves of a SIMD (w and w + 4, probes/simd_map_probe.cpp) run the
// SAME instruction stream half a step apart: waves 0-3 pass a phase's barrier BEHIND its dx = 0 block, waves 4-7 IN FRONT of it --
// released together, one starts with its piece and the other with its block, and they alternate until the next barrier.
// vmcnt (gfx9 counts loads AND stores in it, in order): a phase issues chunk ph + 2 FIRST and its slot's stores / loads AFTER it,
// and the wait in front of the next barrier is vmcnt(N), N = a LOWER BOUND of the memory instructions that slot has issued
// (18 / 16 -> 15, what the 4-bit field takes; 2; else 0) -- the chunk has landed, the slot's traffic may stay in flight for a
// second phase.  dy = 0 phases carry no slot: there the 5 input pieces are the youngest and vmcnt(5) at dy = 1 is as before.
// hipcc: the slot code must not share a LOOP with the residual loads of an earlier trip, nor sit in a wave-group branch: it then
// assumes those rows are still in flight when a slot touches the registers and waits with ITS count (blind to the LDS-DMA) --
// vmcnt(0) in the middle of a slot (profiles/r5a, r5b).  Hence the separate unrolled slot trip and one stream for all waves.
//
// Upstream call sites replaced: conv1 / conv2 tasks of scripts/tilevae.py:115-136 with the custom_group_norm + SiLU in front of
// the NEXT conv (:218-245, :102-104) and the queue's add_res (:612-616) -- the same set as vae_conv_rec.hip.
#include "common.h"

using namespace mdt;

#include "conv_rec_common.h"

namespace {

struct DItem {
    int b, cb, y0, x0;      // cb: 64-cout block
};

constexpr int D_MT = 2;                  // 32-cout tiles of an item (and of a wave)
constexpr int D_NROW = 2;                // pixel rows of a wave
constexpr int D_TH = 16, D_ROWS = D_TH + 2, D_COLS = 34;
constexpr int ECD = 16;                  // float4 stride between [bias | a | s] of a constants buffer (64 couts x 4 B = 256 B each)
constexpr int ECD_REC = 3 * ECD;         // records of one constants buffer

constexpr int D_TK = 4;                  // K-steps of the slot trip
// slot of phase p (0 .. 11) of an item's first trip: position i = 2 k + dy - 1 of the dy = 1 / 2 phases, 8 slots
__host__ __device__ constexpr int slot_pos(int p) { return p % 3 == 0 ? -1 : 2 * (p / 3) + p % 3 - 1; }
__host__ __device__ constexpr int slot_kind(int p) { return slot_pos(p) < 0 ? 0 : 1 + slot_pos(p) % 2; }   // 0 none, 1: A + R0, 2: R1 + E
__host__ __device__ constexpr int slot_unit(int p) { return slot_pos(p) / 2; }

#define MDT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))      // s_waitcnt vmcnt(n), n <= 15 (expcnt / lgkmcnt untouched)

__global__ __launch_bounds__(512, 2) void k_conv3x3_recd(const ConvRParams P) {
    using IS = InStage<D_ROWS>;
    constexpr int W_REC = 2 * 3 * D_MT * 64;          // records of a chunk in LDS  [hl][dx][mt][lane]
    constexpr int W_SRC = 2 * 3 * 4 * 64;             // records of the packed chunk in HBM  [hl][dx][mt 4][lane] (128-cout blocks)
    constexpr int W_DMA = W_REC / 64;                 // 12 pieces: waves 0-7 take piece w, waves 0-3 also piece w + 8
    static_assert(IS::DMA % 8 == 0 && IS::PW == 5 && W_DMA == 12, "the counted wait assumes 5 input pieces per wave");
    __shared__ u32x4 smem[2 * IS::PAD + 3 * W_REC + 2 * ECD_REC];
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + 2 * IS::PAD;
    u32x4* const ec_l = smem + 2 * IS::PAD + 3 * W_REC;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hp = P.H + 2, Wp = rec_pitch(P.W), Pn = P.Cin >> 3, PnO = P.Cout >> 3;
    const size_t plane = (size_t)Hp * Wp;
    const int NCB2 = P.Cout >> 6;
    // probing (PROBES twin only, MDTILE_REC_DBG): 1 K loop alone | 2 no activation arithmetic | 4 no record stores | 8 no fp32 stores |
    // 16 no residual loads | 32 the barriers do not wait for the DMA (wrong operands, timing only) | 64 no operand DMA at all
    const int dbg = pdbg(P.dbg);

    // work -> (sample, pixel tile, 64-cout block): as in k_conv3x3_rec (`work % 8` is this block's XCD for all its items: every cout
    // block of a pixel tile stays on one L2)
    const int per_img = ((P.ptiles + 7) / 8) * 8 * NCB2, total = per_img * P.B;
    auto decode = [&](int work, DItem& it) -> bool {
        it.b = work / per_img;
        const int r = work - it.b * per_img, xcd = r & 7, slot = r >> 3;
        const int ptile = (slot / NCB2) * 8 + xcd;
        it.cb = slot % NCB2;
        const int py = ptile / P.PX, px = ptile - py * P.PX;
        it.y0 = py * D_TH;
        it.x0 = px * 32;
        return ptile < P.ptiles;
    };
    auto next_valid = [&](int work, DItem& it) -> int {
        while (work < total && !decode(work, it)) work += gridDim.x;
        return work;
    };

    auto make_ioff = [&](const DItem& it, unsigned (&ioff)[IS::PW]) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            const int di = wave + 8 * i;
            int s = (di % IS::HALF_DMA) * 64 + ln;
            if (s >= IS::HALF) s = IS::HALF - 1;
            const int g = s / (D_ROWS * D_COLS), p = s - g * (D_ROWS * D_COLS);
            const int r = p / D_COLS, c = p - r * D_COLS;
            int pr = it.y0 + r, pc = it.x0 + c;
            pr = pr < Hp ? pr : Hp - 1;
            pc = (pc < P.W + 1 ? pc : P.W + 1) + REC_COL0;
            ioff[i] = (unsigned)(((size_t)g * plane + (size_t)pr * Wp + pc) * 16);
        }
    };
    // ---- operand stream addresses.  A DMA piece costs the wave ~100 cycles of issue as it is (probes/conv_drip_ab.py --dbg 97: the K loop
    // without its 88 pieces per item is 10 % faster); hipcc's address arithmetic per piece -- a 64-bit multiply by the plane stride, SGPR
    // spill reloads -- made it 18 instructions.  The sources are therefore RUNNING pointers (SGPR pairs advanced by additions only, opaque
    // to the optimiser so that they are never re-derived from k), the per-wave parts sit in lane offsets / per-wave LDS addresses:
    //   input : three pointers (pieces of the hi half | piece 2, whose half depends on the wave | pieces of the lo half), + 2 planes per K-step
    //   chunks: one pointer, + one packed chunk (24 KB) per phase; the wave's piece offset inside the chunk is part of the lane offset
    struct P64 {      // a 64-bit address as two scalar registers (hipcc moves a 64-bit value it ADDS to into vector registers: no s_add_u64 on gfx9)
        unsigned lo, hi;
    };
    auto p64 = [](const void* q) {
        const size_t v = reinterpret_cast<size_t>(q);
        P64 r;
        r.lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        r.hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        asm volatile("" : "+s"(r.lo), "+s"(r.hi));
        return r;
    };
    auto add64 = [](P64& a, const P64& k) { asm volatile("s_add_u32 %0, %0, %2\n\ts_addc_u32 %1, %1, %3" : "+s"(a.lo), "+s"(a.hi) : "s"(k.lo), "s"(k.hi) : "scc"); };
    auto ptr_of = [](const P64& a) { return reinterpret_cast<const char*>(((size_t)a.hi << 32) | a.lo); };
    auto opaque = [&](const char* q) { return ptr_of(p64(q)); };
    auto lds_addr = [](const u32x4* q) { return (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) u32x4*)q; };
    const size_t hloff = (size_t)Pn * plane * 16;
    const P64 kstride = p64(reinterpret_cast<const void*>((size_t)2 * plane * 16));
    const bool mid_lo = (wave + 16) / IS::HALF_DMA != 0;      // piece 2 = DMA wave + 16: record half (wave + 16) / 20
    P64 x_hi = {0u, 0u}, x_mid = {0u, 0u}, x_lo = {0u, 0u};      // (named by record half: hi = hl 0)
    auto input_at = [&](const DItem& it) {      // K-step 0 of an item
        const char* xb = reinterpret_cast<const char*>(P.x + (size_t)it.b * 2 * Pn * plane);
        x_hi = p64(xb);
        x_lo = p64(xb + hloff);
        x_mid = p64(mid_lo ? xb + hloff : xb);
    };
    auto input_next = [&]() {
        add64(x_hi, kstride);
        add64(x_lo, kstride);
        add64(x_mid, kstride);
    };
    const unsigned in_lds_w = lds_addr(in_l) + (unsigned)wave * 1024u;      // + stage * PAD * 16 + i * 8192
    auto issue_input = [&](const unsigned (&ioff)[IS::PW], int stage, int i0 = 0, int i1 = IS::PW) {      // the K-step the pointers stand at
        if (dbg & 64) return;
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            if (i < i0 || i >= i1) continue;
            dma16(ptr_of(i < 2 ? x_hi : (i == 2 ? x_mid : x_lo)), ioff[i], reinterpret_cast<const u32x4*>(0), in_lds_w + (unsigned)(stage * IS::PAD * 16 + i * 8192));
        }
    };
    const unsigned lane16 = lane * 16;
    // chunk (k, dy) of the item's 64 couts: piece p = (hl, dx, mt) -> packed piece (hl, dx, 2 (cb & 1) + mt) of the 128-cout block cb >> 1
    P64 wptr = {0u, 0u};             // the next chunk to request
    const P64 wstride = p64(reinterpret_cast<const void*>((size_t)W_SRC * 16));
    unsigned wvoff[2], w_lds_w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pw = wave + 8 * i;
        wvoff[i] = lane16 + (unsigned)(((pw >> 1) * 4 + (pw & 1)) * 1024);
        w_lds_w[i] = lds_addr(w_l) + (unsigned)pw * 1024u;      // + ring * W_REC * 16
    }
    auto weights_at = [&](const DItem& it) {      // chunk 0 of an item
        wptr = p64(reinterpret_cast<const char*>(P.w + (size_t)(it.cb >> 1) * P.NK * 3 * W_SRC) + (it.cb & 1) * 2048);
    };
    auto issue_weights = [&](int ring) {          // the chunk the pointer stands at; then on to the next one
        if (!(dbg & 64)) {
            dma16(ptr_of(wptr), wvoff[0], reinterpret_cast<const u32x4*>(0), w_lds_w[0] + (unsigned)(ring * W_REC * 16));
            if (wave < 4) dma16(ptr_of(wptr), wvoff[1], reinterpret_cast<const u32x4*>(0), w_lds_w[1] + (unsigned)(ring * W_REC * 16));
        }
        add64(wptr, wstride);
    };
    // epilogue constants of an item's 64 couts: waves 0 / 1 / 2 fetch bias / a / s, 256 B each (lanes 0-15)
    auto issue_consts = [&](const DItem& it, int par) {
        if (lane < 16) {
            if (wave == 0 && P.bias) dma16(reinterpret_cast<const char*>(P.bias + it.cb * 64), lane16, ec_l + par * ECD_REC);
            if ((wave == 1 || wave == 2) && P.yrec && P.coef)
                dma16(reinterpret_cast<const char*>(P.coef + ((size_t)it.b * 2 + (wave - 1)) * P.Cout + it.cb * 64), lane16,
                      ec_l + par * ECD_REC + wave * ECD);
        }
    };

    bf16x8 fw[2][D_MT][2];     // [set][m][hl]
    bf16x8 fx[2][D_NROW][2];   // [set][row][hl]
    const int wfrag = lane;                                          // + ((hl*3 + dx)*2 + m)*64
    const int xfrag = (kg * D_ROWS + wave * D_NROW) * D_COLS + l31;  // + hl*HALF_PAD + (n + dy)*COLS + dx
    auto load_fw = [&](int set, int ring, int dx) {
        const u32x4* wst = w_l + ring * W_REC + wfrag;
#pragma unroll
        for (int m = 0; m < D_MT; ++m)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fw[set][m][hl] = __builtin_bit_cast(bf16x8, wst[((hl * 3 + dx) * D_MT + m) * 64]);
    };
    auto load_fx = [&](int set, int stage, int dy, int dx) {
        const u32x4* ist = in_l + stage * IS::PAD + xfrag + dy * D_COLS + dx;
#pragma unroll
        for (int n = 0; n < D_NROW; ++n)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fx[set][n][hl] = __builtin_bit_cast(bf16x8, ist[hl * IS::HALF_PAD + n * D_COLS]);
    };

    // ------------------------------------------------------------------------------------------------------------------
    // the epilogue of unit u = (m-tile, row) of a finished item, in the pieces the slots issue (addressing as epilogue_item:
    // wave-uniform 64-bit base in SGPRs + 32-bit lane offset; constants read from LDS where they are used)
    const size_t HW = (size_t)P.H * P.W, HW4 = HW * sizeof(float);
    const size_t pl16 = plane * sizeof(u32x4), lo_half = (size_t)PnO * pl16;
    const bool has_act = P.yrec != nullptr && P.coef != nullptr;
    struct Lane {
        int x, y;
        bool ok;
        unsigned kgo, xc;
    };
    auto lane_of = [&](const DItem& it, int n) {
        Lane L;
        unsigned kgo = (unsigned)kg;
        asm volatile("" : "+v"(kgo));      // formed per slot: as loop invariants these offsets end up in scratch
        int le = lane;
        asm volatile("" : "+v"(le));
        L.kgo = kgo;
        L.x = it.x0 + (le & 31);
        L.y = it.y0 + wave * D_NROW + n;
        L.ok = L.x < P.W && L.y < P.H;
        L.xc = (unsigned)(L.x < P.W ? L.x : 0);
        return L;
    };
    // Output addresses of a finished item: ONE 64-bit base per tensor and item (SGPR pair, formed when the item ends) + a 32-bit lane offset
    // that carries the m-tile, the lane's half, the pixel AND the plane walk (v_add per access).  epilogue_item steps a scalar pointer per
    // plane instead (2 SALU per access): fine with idle matrix cores, but next to the MFMAs of the partner wave a scalar instruction is the
    // expensive kind (profiles/r5g: 1.6 cycles of matrix pipe per SALU, 0.7 per VALU; this kernel ran 1.74 SALU per MFMA).
    // 32-bit offsets: 64 planes x HW x 4 B and 8 planes x (H + 2) x pitch x 16 B stay below 2^32 (rec_image_ok on the host).
    struct Bases {
        gchar *y32, *rec_hi, *rec_lo;
    };
    auto opaque_g = [&](const void* q) { return (gchar*)reinterpret_cast<size_t>(opaque(reinterpret_cast<const char*>(q))); };
    auto bases_of = [&](const DItem& it) {
        Bases Bs;
        Bs.y32 = opaque_g(reinterpret_cast<const char*>(P.y32) + ((size_t)it.b * P.Cout + (size_t)it.cb * 64) * HW4);
        const char* rh = reinterpret_cast<const char*>(P.yrec) + ((size_t)it.b * 2 * PnO + (size_t)it.cb * 8) * pl16;
        Bs.rec_hi = opaque_g(rh);
        Bs.rec_lo = opaque_g(rh + lo_half);
        return Bs;
    };
    const unsigned hw4u = (unsigned)HW4, hw4u5 = 5u * (unsigned)HW4;
    // A: + bias, fp32 stores
    auto slot_a = [&](f32x16 (&S)[D_MT][D_NROW][1], const DItem& it, const Bases& Bs, const u32x4* ec, int m, int n) {
        const Lane L = lane_of(it, n);
        if (P.bias) {
            const float4* e4 = reinterpret_cast<const float4*>(ec) + (m * 8 + L.kgo);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 tb = e4[2 * g];
                S[m][n][0][4 * g] += tb.x; S[m][n][0][4 * g + 1] += tb.y; S[m][n][0][4 * g + 2] += tb.z; S[m][n][0][4 * g + 3] += tb.w;
            }
        }
        if (P.y32 && L.ok && !(dbg & 8)) {
            unsigned off = (((unsigned)(m * 32) + 4u * L.kgo) * (unsigned)HW + (unsigned)(L.y * P.W) + L.xc) * 4u;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q) off += MDT_PLANE_STEP(q) == 1 ? hw4u : hw4u5;
                *(MDT_GLOBAL float*)(Bs.y32 + (size_t)off) = S[m][n][0][q];
            }
        }
    };
    // R: activation (in place), then split + record stores, of 8 of the lane's 16 channels (R = 0 / 1) -- two pieces of a slot
    auto slot_r_act = [&](f32x16 (&S)[D_MT][D_NROW][1], const u32x4* ec, int m, int n, int R) {
        if (!has_act || (dbg & 2)) return;
        unsigned kgo = (unsigned)kg;
        asm volatile("" : "+v"(kgo));
        const float4* e4 = reinterpret_cast<const float4*>(ec) + (m * 8 + kgo);
        f32x2 aq[4], sq[4];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float4 ta = e4[ECD + 4 * R + 2 * g], ts = e4[2 * ECD + 4 * R + 2 * g];
            aq[2 * g] = f32x2{ta.x, ta.y}; aq[2 * g + 1] = f32x2{ta.z, ta.w};
            sq[2 * g] = f32x2{ts.x, ts.y}; sq[2 * g + 1] = f32x2{ts.z, ts.w};
        }
        act8(S[m][n][0], 8 * R, aq, sq);
    };
    auto slot_r_store = [&](f32x16 (&S)[D_MT][D_NROW][1], const DItem& it, const Bases& Bs, int m, int n, int R) {
        if (!P.yrec) return;
        const Lane L = lane_of(it, n);
        if (!L.ok) return;
        // record (row y + 1, padded column x + 1) of plane (m 4 + 2 R + kg) of the item's 8
        const size_t at = (size_t)((((unsigned)(m * 4 + 2 * R) + L.kgo) * (unsigned)plane + (unsigned)((L.y + 1) * Wp) + L.xc + (unsigned)(REC_COL0 + 1)) * 16u);
        u32x4 hi, lo;
        split8p(S[m][n][0], 8 * R, hi, lo);
        if (!(dbg & 4)) {
            *(MDT_GLOBAL u32x4*)(Bs.rec_hi + at) = hi;
            *(MDT_GLOBAL u32x4*)(Bs.rec_lo + at) = lo;
        } else {
            asm volatile("" ::"v"(hi), "v"(lo));      // (probing: the split stays, its stores go)
        }
    };
    // zero border of the record image: this block owns the border cells next to its edge pixels.  Coordinates only, no accumulator data:
    // ONE rolled loop per item over the wave's 2 rows x 2 m-tiles x 2 record pairs (edge tiles only; issued with the item's last R slot)
    auto zero_border = [&](const DItem& it) {
        if (!P.yrec) return;
        if (!(it.x0 == 0 || it.x0 + 32 >= P.W || it.y0 == 0 || it.y0 + D_TH >= P.H)) return;      // (wave-uniform: interior tiles own no border cell)
        const Lane L0 = lane_of(it, 0);
        if (!(L0.x < P.W)) return;
        const bool left = L0.x == 0, right = L0.x + 1 == P.W;
        const u32x4 z = {0u, 0u, 0u, 0u};
        const unsigned p0 = (L0.kgo * (unsigned)plane + (unsigned)REC_COL0) * 16u;
#pragma unroll 1
        for (int j = 0; j < 2 * D_NROW * D_MT; ++j) {
            const int n = j & 1, RR = (j >> 1) & 1, m = j >> 2;
            const int y = L0.y + n;
            if (y >= P.H) continue;
            const bool top = y == 0, bot = y == P.H - 1;
            if (!(left || right || top || bot)) continue;
            char* const ypz = reinterpret_cast<char*>(P.yrec) + ((size_t)it.b * 2 * PnO + (size_t)(it.cb * D_MT + m) * 4 + 2 * RR) * pl16;
            auto zrec = [&](int py, int px) {      // (px: padded column, 0 = left border)
                const size_t az = (size_t)(p0 + (unsigned)(py * Wp + px) * 16u);
                *reinterpret_cast<u32x4*>(ypz + az) = z;
                *reinterpret_cast<u32x4*>(ypz + lo_half + az) = z;
            };
            if (left) zrec(y + 1, 0);
            if (right) zrec(y + 1, P.W + 1);
            if (top) {
                zrec(0, L0.x + 1);
                if (left) zrec(0, 0);
                if (right) zrec(0, P.W + 1);
            }
            if (bot) {
                zrec(P.H + 1, L0.x + 1);
                if (left) zrec(P.H + 1, 0);
                if (right) zrec(P.H + 1, P.W + 1);
            }
        }
    };
    // E: the unit's registers are stored -- the residual row of the block's NEXT item goes into them
    // (loads only: without a residual the start values are zeroed at the item top -- a zeroing `else` here made hipcc merge the two
    // paths with v_cndmask behind a vmcnt(0) of its own, i.e. wait for the rows right where they were requested)
    const bool res_on = P.res != nullptr && !(dbg & 1) && !(dbg & 16);
    auto slot_e = [&](f32x16 (&S)[D_MT][D_NROW][1], const DItem& it, gchar* resb, bool on, int m, int n) {
        if (res_on && on) {
            // (clamped coordinates: unconditional loads, as residual_into_acc)
            unsigned kgo = (unsigned)kg;
            asm volatile("" : "+v"(kgo));
            int le = lane;
            asm volatile("" : "+v"(le));
            const int x = it.x0 + (le & 31), y = it.y0 + wave * D_NROW + n;
            const unsigned xc = (unsigned)(x < P.W ? x : 0), yc = (unsigned)(y < P.H ? y : P.H - 1);
            unsigned off = (((unsigned)(m * 32) + 4u * kgo) * (unsigned)HW + yc * (unsigned)P.W + xc) * 4u;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q) off += MDT_PLANE_STEP(q) == 1 ? hw4u : hw4u5;
                S[m][n][0][q] = *(const MDT_GLOBAL float*)(resb + (size_t)off);
            }
        }
    };
    auto res_base_of = [&](const DItem& it) { return opaque_g(reinterpret_cast<const char*>(P.res) + ((size_t)it.b * P.Cout + (size_t)it.cb * 64) * HW4); };

    // ------------------------------------------------------------------------------------------------------------------
    DItem cur, nxt, prv;
    int work = next_valid(blockIdx.x, cur);
    if (work >= total) return;
    prv = cur;
    unsigned ioff[IS::PW];
    make_ioff(cur, ioff);
    input_at(cur);
    issue_input(ioff, 0);
    input_next();
    weights_at(cur);
    issue_weights(0);
    issue_weights(1);
    issue_consts(cur, 0);
    const int nph = P.NK * 3;
    int par = 0;                 // constants buffer of `cur`; the sealed item's is par ^ 1
    bool have_prev = false;
    Bases pb = bases_of(cur);      // output bases of `prv`

    f32x16 acc[D_MT][D_NROW][1], sealed[D_MT][D_NROW][1];
    // the first item's start values go into `sealed` and change sides at the loop top like every later item's
#pragma unroll
    for (int m = 0; m < D_MT; ++m)
#pragma unroll
        for (int n = 0; n < D_NROW; ++n) {
            slot_e(sealed, cur, res_base_of(cur), true, m, n);
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][n][0][q] = 0.0f;
            if (!res_on) {
#pragma unroll
                for (int q = 0; q < 16; ++q) sealed[m][n][0][q] = 0.0f;
            }
        }

    while (true) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of the item's first operands have landed, its residual rows, and every store of the slots is out
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the sets change roles: acc <- start values of `cur` (residual or zeros), sealed <- the sums of `prv`
#pragma unroll
        for (int m = 0; m < D_MT; ++m)
#pragma unroll
            for (int n = 0; n < D_NROW; ++n) {
                const f32x16 t = acc[m][n][0];
                acc[m][n][0] = sealed[m][n][0];
                sealed[m][n][0] = t;
            }
        if (!res_on) {
#pragma unroll
            for (int m = 0; m < D_MT; ++m)
#pragma unroll
                for (int n = 0; n < D_NROW; ++n)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[m][n][0][q] = 0.0f;
        }
        load_fw(0, 0, 0);
        load_fx(0, 0, 0, 0);
        const int work_n = next_valid(work + gridDim.x, nxt);
        const bool has_next = work_n < total;
        const u32x4* const ec_prev = ec_l + (par ^ 1) * ECD_REC;
        gchar* const resb = res_base_of(nxt);
        const bool drip = have_prev && !(dbg & 1);

        // the requests of a phase in three pieces, one behind each of its MFMA blocks (see the file header)
        //   piece 0: chunk ph+2, then the first third of EITHER the slot (dy = 1, 2 of the slot trip) OR the next K-step's input stage
        //   (dy = 0); pieces 1, 2: the rest of it.  Per wave the order is chunk -> slot / input, all in front of the next phase's wait.
        auto piece = [&](int sp, int qdy, int kq, int j, bool can_be_last) {      // sp: phase of the slot trip (0 .. 11) or -1, tap row, K-step, piece j
            const int qph = kq * 3 + qdy;
            const int sk = sp >= 0 ? slot_kind(sp) : 0, su = sk ? slot_unit(sp) : 0, sm = su >> 1, sn = su & 1;
            if (j == 0 && qph + 2 < nph) issue_weights((qdy + 2) % 3);
            if (sk == 1 && drip) {
                if (j == 0) slot_a(sealed, prv, pb, ec_prev, sm, sn);
                else if (j == 1) slot_r_act(sealed, ec_prev, sm, sn, 0);
                else slot_r_store(sealed, prv, pb, sm, sn, 0);
            }
            if (sk == 2) {
                if (j == 0) { if (drip) slot_r_act(sealed, ec_prev, sm, sn, 1); }
                else if (j == 1) {
                    if (drip) {
                        slot_r_store(sealed, prv, pb, sm, sn, 1);
                        if (su == 3) zero_border(prv);
                    }
                } else slot_e(sealed, nxt, resb, has_next, sm, sn);      // the residual row of the block's next item into the unit just stored from
            }
            if (qdy == 0 && kq + 1 < P.NK) {
                issue_input(ioff, (kq + 1) & 1, 2 * j, j == 2 ? IS::PW : 2 * j + 2);
                if (j == 2) input_next();
            }
            if (j == 0 && qdy == 2 && can_be_last && kq + 1 == P.NK && has_next) {      // (can_be_last: second K-step of a plain trip, compile time)
                // last phase of the item: ring slots 0 / 1 and input stage 0 are out of use -> the next item's first operands
                // (its lane offsets are formed here, not at the item top: 5 registers less across the K loop)
                make_ioff(nxt, ioff);
                input_at(nxt);
                issue_input(ioff, 0);
                input_next();
                weights_at(nxt);
                issue_weights(0);
                issue_weights(1);
                issue_consts(nxt, par ^ 1);
            }
        };
        // the wait in front of a phase's barrier.  In flight, oldest first: [slot traffic of phase ph-2] chunk ph+1 (requested behind
        // the previous barrier), then EITHER the slot traffic of phase ph-1 (a lower bound N of its memory instructions is known, see
        // the file header) OR -- at dy = 1 -- the 5 input pieces of K-step k+1: those youngest ones may stay in flight another phase.
        auto phase_wait = [&](int psp, int dy, int k) {          // psp: slot-trip phase of the PREVIOUS phase, or -1
            const int pk = psp >= 0 ? slot_kind(psp) : 0;
            bool counted = (dbg & 32) != 0;                      // (probing: no wait at all)
            if (pk != 0 && dbg == 0) {                           // (the other probing switches change what a slot issues: plain vmcnt(0) then)
                const bool yok = drip && prv.y0 + wave * D_NROW + (slot_unit(psp) & 1) < P.H;      // wave-uniform: did the slot issue its stores
                const bool big = pk == 1 ? (yok && P.y32 != nullptr) : (res_on && has_next);       // >= 16 of them
                if (big) { MDT_VMCNT(15); counted = true; }
                else if (yok && P.yrec != nullptr) { MDT_VMCNT(2); counted = true; }
            }
            if (!counted) {
                if (dy == 1 && k + 1 < P.NK) MDT_VMCNT(5);
                else MDT_VMCNT(0);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // one step = [waves 4-7: the phase's barrier] fragments of the next step out of LDS, 12 MFMAs, [waves 0-3: the phase's barrier], piece dx
        //   t: step of the unrolled trip, T: its length, kb: first K-step of the trip, SLOTS: the slot trip (the item's K-steps 0 .. 3)
        auto step = [&](int t, int T, int kb, bool SLOTS, bool more) {
            const int kk = t / 9, dy = (t / 3) % 3, dx = t % 3, pl_ = t / 3;
            const int k = kb + kk;
            const int ws = t & 1;
            // the phase before this one as a slot-trip phase: of this trip, or (first phase of the first plain trip: kb == D_TK, a run-time
            // test) the slot trip's last one; the item's very first phase has none (the item top waited for everything)
            auto wait_here = [&]() {
                if (!SLOTS && pl_ == 0) {
                    if (kb == D_TK) phase_wait(3 * D_TK - 1, dy, k);
                    else phase_wait(-1, dy, k);
                } else {
                    phase_wait(SLOTS ? pl_ - 1 : -1, dy, k);
                }
            };
            if (dx == 0 && wave >= 4) wait_here();
            MDT_PIN();
            // (one explicit LDS wait per block: everything but the 8 fragment reads just issued -- hipcc otherwise stages four of its own, lgkmcnt(11) .. (8))
            if (t < T - 1) {
                const int t1 = t + 1;
                load_fw(ws ^ 1, (t1 / 3) % 3, t1 % 3);
                load_fx(ws ^ 1, (t1 / 9) & 1, (t1 / 3) % 3, t1 % 3);
                __builtin_amdgcn_s_waitcnt(0xC87F);      // lgkmcnt(8)
            } else if (more) {
                load_fw(ws ^ 1, 0, 0);
                load_fx(ws ^ 1, 0, 0, 0);
                __builtin_amdgcn_s_waitcnt(0xC87F);
            } else {
                __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
            }
            MDT_PIN();
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int n = 0; n < D_NROW; ++n)
#pragma unroll
                    for (int m = 0; m < D_MT; ++m)
                        acc[m][n][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ws][m][term == 0 ? 1 : 0], fx[ws][n][term == 1 ? 1 : 0],
                                                                               acc[m][n][0], 0, 0, 0);   // w_lo x_hi, w_hi x_lo, w_hi x_hi
            MDT_PIN();
            if (dx == 0 && wave < 4) wait_here();
            piece(SLOTS ? pl_ : -1, dy, k, dx, !SLOTS && kk == 1);
        };

        // the slot trip: K-steps 0 .. 3 of the item, the previous item's epilogue and the next item's residual rows in its slots
#pragma unroll
        for (int t = 0; t < 9 * D_TK; ++t) step(t, 9 * D_TK, 0, true, true);
        // the rest of the K loop in trips of two K-steps (its own unrolled body, see the file header)
        for (int k2 = D_TK; k2 < P.NK; k2 += 2) {
#pragma unroll
            for (int t = 0; t < 18; ++t) step(t, 18, k2, false, k2 + 2 < P.NK);
        }

        prv = cur;
        pb = bases_of(cur);
        have_prev = true;
        if (!has_next) break;
        work = work_n;
        cur = nxt;
        par ^= 1;
    }

    // the block's last item: nothing left to hide under -- its epilogue in one piece
    if (!(dbg & 1)) {
        const u32x4* const ec_last = ec_l + par * ECD_REC;
#pragma unroll
        for (int m = 0; m < D_MT; ++m)
#pragma unroll
            for (int n = 0; n < D_NROW; ++n) {
                slot_a(acc, prv, pb, ec_last, m, n);
                slot_r_act(acc, ec_last, m, n, 0);
                slot_r_store(acc, prv, pb, m, n, 0);
                slot_r_act(acc, ec_last, m, n, 1);
                slot_r_store(acc, prv, pb, m, n, 1);
            }
        zero_border(prv);
    }
}

}  // namespace

namespace mdt {

// cin % 64 == 0 (whole 4-K-step trips), 128-cout packed blocks; H, W: output = input size
bool conv_recd_supported(int cout, int cin) { return cin % 32 == 0 && cin >= 128 && cout % 128 == 0; }

int conv_recd_launch(ConvRParams P, int B, hipStream_t s, int cus) {
    P.PX = (P.W + 31) / 32;
    P.ptiles = P.PX * ((P.H + 15) / 16);
    const long long items = (long long)((P.ptiles + 7) / 8) * 8 * (P.Cout / 64) * B;
    const int grid_max = cus / 8 * 8;
    dim3 grid((unsigned)(items < grid_max ? items : grid_max)), block(512);
    hipLaunchKernelGGL(k_conv3x3_recd, grid, block, 0, s, P);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

}  // namespace mdt
