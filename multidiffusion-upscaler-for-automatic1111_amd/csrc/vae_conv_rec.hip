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

#define MDT_REC_KERNEL k_conv3x3_rec
#define MDT_REC_ST 0
#include "vae_conv_rec_direct_body.h"
#undef MDT_REC_KERNEL
#undef MDT_REC_ST
#define MDT_REC_KERNEL k_conv3x3_rec_st
#define MDT_REC_ST 1
#include "vae_conv_rec_direct_body.h"
#undef MDT_REC_KERNEL
#undef MDT_REC_ST

// =====================================================================================================================
// nearest-2x upsample + 3x3 conv in sub-pixel form (four 2x2 convs on the un-upsampled grid, see vae_conv_bf16x3.hip
// k_upconv_bf16x3 for the derivation).  Item = 128 couts x (8 x 32 INPUT px) of ONE output-row parity a and both column
// parities bb; phases (K-step k, tap row u); a phase = 4 combo-steps c of 12 MFMAs per wave:
//     c:  0 (shift s 0, bb 0)   1 (s 1, bb 0)   2 (s 1, bb 1)   3 (s 2, bb 1)        tap column v = s - bb
// Weight chunk [hl][bb][v][mt][lane] per (a, cb, k, u) in a 3-slot ring.  Persistent like k_conv3x3_rec; an item has 2 NK
// phases, which need not be a multiple of 3, so the ring position of an item's first chunk (r0) rotates from item to item:
// the next item's chunks 0 / 1 go to the two slots the last phase is NOT reading.
#define MDT_REC_KERNEL k_upconv_rec
#define MDT_REC_ST 0
#include "vae_conv_rec_upconv_body.h"
#undef MDT_REC_KERNEL
#undef MDT_REC_ST
#define MDT_REC_KERNEL k_upconv_rec_st
#define MDT_REC_ST 1
#include "vae_conv_rec_upconv_body.h"
#undef MDT_REC_KERNEL
#undef MDT_REC_ST

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
// probes/csrc/vae_conv_recd.hip (PROBES twin of the library only -- a measured, rejected form: DESIGN.md / docs/history/r5.md): 64-cout items with
// the epilogue dripped into the next item's K loop (direct 3x3, cin % 32 == 0, cin >= 128).  Declared here, defined only in that build: the one
// call sits in a discarded `if constexpr (kProbes)` statement, so the shipping library neither references nor carries the kernel.
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
// statistics in the epilogue: one-block-per-CU kernels of the 128-cout family only.  A launch the cost model would hand to the two-blocks
// family (a few item rounds: the 278 x 278 level of the 8K decode) gains more from that family than from dropping the statistics pass.
bool conv_rec_stats_in_epilogue(int B, int cin, int cout, int H, int W, int up) {
    if (cout % 128 != 0 || !rec_persistent()) return false;
    const int cus = num_cus() / 8 * 8;
    const int hin = up ? H / 2 : H, win = up ? W / 2 : W, per = (cout / 128) * (up ? 2 : 1);
    const long long px = (win + 31) / 32;
    const long long items16 = (px * ((hin + (up ? 7 : 15)) / (up ? 8 : 16)) + 7) / 8 * 8 * per * B;
    const long long items8 = (px * ((hin + (up ? 3 : 7)) / (up ? 4 : 8)) + 7) / 8 * 8 * per * B;
    (void)cin;
    return !rec_two_blocks(items16, items8, cus, up, 0);
}
// units of the per-wave partials a statistics launch writes (conv_stats_finish_launch): [B][units][NCB][32 quads][2] doubles
int conv_rec_stats_units(int H, int W, int up) {
    return up ? ((W / 2 + 31) / 32) * ((H / 2 + 7) / 8) * 8 : ((W + 31) / 32) * ((H + 15) / 16) * 4;
}

int conv_rec_launch(const void* d_xrec, const void* d_w_rec, const float* d_bias, const float* d_res, float* d_y32, void* d_yrec,
                    const float* d_ycoef, int B, int cin, int cout, int H, int W, int up, hipStream_t s, const int* win, int family,
                    double* d_part) {
    ConvRParams P;
    P.gn_part = d_part;
    if (d_part) {
        MDT_CHECK_ARG(cout % 128 == 0 && rec_persistent(), "conv_rec_launch: no statistics kernel for cout=%d", cout);
        family = 1;
    }
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
    if constexpr (kProbes) {
        if (family == 3) {      // a NAMED family is honoured or refused, never silently replaced by the cost model
            MDT_CHECK_ARG(!up && !win && !d_part && conv_recd_supported(cout, cin), "conv_rec_launch: the dripped-epilogue kernel takes direct 3x3 convs with cin %% 32 == 0, "
                          "cin >= 128, cout %% 128 == 0, no window, no statistics (cout=%d cin=%d up=%d)", cout, cin, up);
            return conv_recd_launch(P, B, s, num_cus());
        }
    }
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
        if (d_part) hipLaunchKernelGGL(k_upconv_rec_st, grid, block, 0, s, P);
        else hipLaunchKernelGGL(k_upconv_rec, grid, block, 0, s, P);
        MDT_LAUNCH_CHECK();
        return MDTILE_OK;
    }
    P.PX = (W + 31) / 32;
    P.ptiles = P.PX * ((H + 15) / 16);
    const long long items = (long long)((P.ptiles + 7) / 8) * 8 * P.NCB * B;
    const int cus = num_cus();                    // one block per CU (155 KB LDS, 2 waves per SIMD)
    if (cout % 128 == 0) stagger(items, cus, (unsigned)P.NK * 900u + 1500u);
    dim3 grid((unsigned)((items < cus || !rec_persistent()) ? items : cus / 8 * 8)), block(512);
    if (d_part) hipLaunchKernelGGL((k_conv3x3_rec_st<2, 2, 4>), grid, block, 0, s, P);
    else if (cout % 128 == 0) hipLaunchKernelGGL((k_conv3x3_rec<2, 2, 4>), grid, block, 0, s, P);
    else hipLaunchKernelGGL((k_conv3x3_rec<1, 1, 2>), grid, block, 0, s, P);      // conv_out: one 32-cout tile, bias padded to 32 by the caller
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

}  // namespace mdt
