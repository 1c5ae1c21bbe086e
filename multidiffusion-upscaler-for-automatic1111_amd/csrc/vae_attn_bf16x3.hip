// Per-tile single-head self-attention of the VAE mid block on the bf16 matrix cores with SPLIT-fp32 operands ("bf16x3"),
// flash style (the T x T score matrix never exists).  Same arithmetic contract as vae_conv_bf16x3.hip: every fp32 factor
// is split x = hi + lo (two bf16, 16 significand bits), a product is x_lo*y_hi + x_hi*y_lo + x_hi*y_hi with fp32
// accumulation inside the MFMA; softmax statistics and the exponentials stay fp32.
// Upstream: tile_utils/attn.py:55-70 (bmm -> softmax -> bmm between the 1x1 convs).
//
// Two kernels:
//   k_attn_prep_*   fp32 q, k ([B,C,T] channel-major) and v ([B,T,C] token-major)  ->  bf16 hi/lo FRAGMENT-ORDER records in
//                   the workspace, padded with zeros to a multiple of 128 tokens, so that the main kernel's LDS stages
//                   are straight 16-byte copies and every MFMA operand is one conflict-free ds_read_b128:
//        Qrec / Krec [B][T128/32 token tiles][C/16 k-steps][hl][lane 64] x 8 bf16 :  token = 32*tile + (lane & 31),
//                                                                                  channel = 16*ks + 8*(lane >> 5) + i
//        Vrec        [B][T128/16 key groups ][C/32 m-tiles][hl][lane 64] x 8 bf16 :  channel = 32*mt + (lane & 31),
//                                                       key = 16*group + (i & 3) + 8*(i >> 2) + 4*(lane >> 5)
//                   The V key order is the order in which a lane of the TRANSPOSED score accumulator holds its keys, so
//                   the probabilities go from the score MFMAs to the output MFMAs without any cross-lane movement.
//   k_attn_bf16x3<C>  block = 512 threads (8 waves), 128 queries; per 128-key block:
//        scores  St[key][query] = sum_c K[key][c] Q[query][c]   A = K (M = keys), B = Q (N = queries), K-dim = channels;
//                wave w owns key tile w >> 1 and the two query tiles of half w & 1; channels stream in 64-channel slabs
//        softmax a lane holds one query column (16 keys of its tile): max / sum in-lane + one 32-lane swap, the 4 key
//                tiles of a query are combined through a few hundred bytes of LDS; P -> bf16 hi/lo records in LDS
//        output  Ot[c][query] += sum_key V[key][c] P[key][query]   A = V^T (M = channels), B = P^T, K-dim = keys;
//                the 128 x C output block lives in registers (8 waves x 8 accumulator tiles for C = 512); the V^T fragments
//                of a wave are its own (nobody shares them) and come straight from global memory
//   LDS: two 64 KB slab stages (K + Q slabs by DMA; the P records of a key block overlay the stage that is free) + statistics.
#include "common.h"

using namespace mdt;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BQ = 128, BK = 128;        // queries per block, keys per iteration
constexpr int P_REC = 8 * 4 * 2 * 64;    // [key k-step 8][query tile 4][hl][lane]
constexpr int STAT_FLOATS = 4 * BQ + 4 * BQ + BQ;   // smax[4][128], ssum[4][128], alpha[128]

__device__ __forceinline__ void split8v(const float (&v)[8], u32x4& hi, u32x4& lo) {
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = (__bf16)v[i];
        l[i] = (__bf16)(v[i] - (float)h[i]);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

// ---- prep: q / k  [C][T] fp32 -> records [tile][ks][hl][lane]
__global__ __launch_bounds__(256) void k_attn_prep_qk(const float* __restrict__ src, u32x4* __restrict__ dst, int C, int T, int tiles) {
    const int NKS = C / 16;
    const size_t n = (size_t)tiles * NKS * 64;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int lane = (int)(idx & 63);
    const int ks = (int)((idx >> 6) % NKS);
    const int tile = (int)((idx >> 6) / NKS);
    const int b = blockIdx.y;
    const int tok = tile * 32 + (lane & 31);
    const float* s = src + (size_t)b * C * T;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = ks * 16 + (lane >> 5) * 8 + i;
        v[i] = tok < T ? s[(size_t)c * T + tok] : 0.0f;
    }
    u32x4 hi, lo;
    split8v(v, hi, lo);
    u32x4* d = dst + ((size_t)b * tiles * NKS + (size_t)tile * NKS + ks) * 128;
    d[lane] = hi;
    d[64 + lane] = lo;
}

// ---- prep: v  [T][C] fp32 token-major -> records [group16][mt][hl][lane]
__global__ __launch_bounds__(256) void k_attn_prep_v(const float* __restrict__ src, u32x4* __restrict__ dst, int C, int T, int groups) {
    const int NMT = C / 32;
    const size_t n = (size_t)groups * NMT * 64;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int lane = (int)(idx & 63);
    const int mt = (int)((idx >> 6) % NMT);
    const int g = (int)((idx >> 6) / NMT);
    const int b = blockIdx.y;
    const int c = mt * 32 + (lane & 31);
    const float* s = src + (size_t)b * T * C;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int key = g * 16 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        v[i] = key < T ? s[(size_t)key * C + c] : 0.0f;
    }
    u32x4 hi, lo;
    split8v(v, hi, lo);
    u32x4* d = dst + ((size_t)b * groups * NMT + (size_t)g * NMT + mt) * 128;
    d[lane] = hi;
    d[64 + lane] = lo;
}

// ---- prep: v  [C][T] fp32 CHANNEL-major (the layout every other 1x1 conv of the queue writes) -> the same records.  A lane owns one
// channel and two runs of 4 consecutive keys: two 16-byte loads when T % 4 == 0 (rows then start 16-byte aligned).
__global__ __launch_bounds__(256) void k_attn_prep_v_cm(const float* __restrict__ src, u32x4* __restrict__ dst, int C, int T, int groups) {
    const int NMT = C / 32;
    const size_t n = (size_t)groups * NMT * 64;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int lane = (int)(idx & 63);
    const int mt = (int)((idx >> 6) % NMT);
    const int g = (int)((idx >> 6) / NMT);
    const int b = blockIdx.y;
    const int c = mt * 32 + (lane & 31);
    const float* s = src + ((size_t)b * C + c) * T;
    float v[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k0 = g * 16 + 8 * h + 4 * (lane >> 5);        // keys k0 .. k0 + 3 = record elements 4 h .. 4 h + 3
        if ((T & 3) == 0) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 < T) t = *reinterpret_cast<const float4*>(s + k0);
            v[4 * h] = t.x; v[4 * h + 1] = t.y; v[4 * h + 2] = t.z; v[4 * h + 3] = t.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * h + i] = k0 + i < T ? s[k0 + i] : 0.0f;
        }
    }
    u32x4 hi, lo;
    split8v(v, hi, lo);
    u32x4* d = dst + ((size_t)b * groups * NMT + (size_t)g * NMT + mt) * 128;
    d[lane] = hi;
    d[64 + lane] = lo;
}

// LDS-DMA of one 16-byte record per lane (global scalar base + 32-bit lane offset -> LDS wave-uniform base + 16 * lane), issued
// through inline asm: hipcc books __builtin_amdgcn_global_load_lds as a pending FLAT access and then turns every later
// `s_waitcnt lgkmcnt(N)` into lgkmcnt(0), which would serialise the fragment prefetch below (same finding as vae_conv_rec.hip).
// Completion is counted by hand: vmcnt(0) + barrier before any ds_read of the data.
__device__ __forceinline__ void dma16a(const void* base, unsigned voff, const u32x4* lds_dst) {
    const unsigned l = (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) u32x4*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(l)
                 : "memory");
}

#define MDT_PIN() __builtin_amdgcn_sched_barrier(0)

// One 128-query block against its key range, per 128-key block:
//   scores   channels stream through LDS in 64-channel slabs [K | Q][tile 4][ks 4][hl][lane] (64 KB) by DMA, two stages; the
//            fragments of k-step ks+1 are read while the MFMAs of ks run (two register sets); ONE barrier per slab (24 MFMAs / wave)
//   softmax  as before; the P records (64 KB) are written INTO the slab stage that is free at that point (no LDS of their own)
//   output   V^T fragments are private to a wave (it owns MT_W channel tiles): they go global -> registers directly, two 16-key
//            steps ahead, never through LDS; P fragments are read one half-step ahead.  No barrier inside the output phase.
// 10 barriers per key block (8 slabs + 2 in the softmax) against 26 of the slab-per-32-channels / V-through-LDS schedule.
template <int C>
__global__ __launch_bounds__(512, 2) void k_attn_bf16x3(const u32x4* __restrict__ Qr, const u32x4* __restrict__ Kr, const u32x4* __restrict__ Vr,
                                                        float* __restrict__ out, int T, int T128, int Tk, int Tk128, float scale, int nsplit,
                                                        float* __restrict__ part, float* __restrict__ pstat) {
    // T / T128: QUERY tokens (rows of the output); Tk / Tk128: KEY tokens.  They differ only when a row band of the queries
    // attends to keys / values gathered from every band (sequence-parallel estimator, mdtile/seqpar.py).
    constexpr int NKS = C / 16, NMT = C / 32, KSS = 4, NSS = NKS / KSS;     // channel k-steps, 32-channel output tiles, k-steps / slab, slabs
    constexpr int WAVES_M = NMT < 8 ? NMT : 8, WAVES_N = 8 / WAVES_M, MT_W = NMT / WAVES_M, NT_W = 4 / WAVES_N;
    constexpr int HALF_REC = 4 * KSS * 2 * 64;                              // K (or Q) part of a slab: [tile 4][ks][hl][lane]
    constexpr int STAGE_REC = 2 * HALF_REC;                                 // 4096 records = 64 KB = exactly the P records of a key block
    static_assert(NSS >= 2 && NSS % 2 == 0 && STAGE_REC == P_REC, "slab sizing");
    __shared__ u32x4 smem[2 * STAGE_REC + STAT_FLOATS / 4];
    u32x4* const slab = smem;
    float* const smax = reinterpret_cast<float*>(smem + 2 * STAGE_REC);
    float* const ssum = smax + 4 * BQ;
    float* const salpha = ssum + 4 * BQ;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kt_w = wave >> 1, qh = wave & 1;                      // score phase: key tile, query-tile pair
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;             // output phase: channel tiles, query tiles
    // block -> (query block, key split), XCD-aware: workgroups go to XCDs round-robin (id % 8), so the `nsplit` key ranges of one
    // query block are handed to consecutive slots of ONE XCD: the Q records every key block re-streams stay in that XCD's L2.
    const int b = blockIdx.y;
    const int bid = blockIdx.x, bxcd = bid & 7, bslot = bid >> 3;
    const int qb = (bslot / nsplit) * 8 + bxcd;
    const int split = bslot - (bslot / nsplit) * nsplit;
    if (qb * BQ >= T128) return;
    const int tiles = T128 / 32, ktiles = Tk128 / 32, groups = Tk128 / 16, nkb = Tk128 / BK;
    const u32x4* Qb = Qr + (size_t)b * tiles * NKS * 128 + (size_t)qb * 4 * NKS * 128;
    const u32x4* Kb = Kr + (size_t)b * ktiles * NKS * 128;
    const u32x4* Vb = Vr + (size_t)b * groups * NMT * 128;

    // slab s of key block kb -> stage: 8 DMA pieces per wave (4 K + 4 Q); piece i covers records [512 (wave/2*... see below)
    // source of tile t, slab s: KSS * 2 * 64 = 512 contiguous records at ((t * NKS + KSS * s) * 128); a wave-instruction moves 64
    // of them: piece index d in [0, 32) of a half -> tile d >> 3, records (d & 7) * 64 .. + 64
    const unsigned lane16 = lane * 16;
    auto issue_S = [&](int kb, int s, int stage) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int half = i >> 2;                     // 0: K, 1: Q (compile-time)
            const int d = wave * 4 + (i & 3);            // 0 .. 31
            const int t4 = d >> 3, off = (d & 7) * 64;
            const u32x4* src = (half == 0 ? Kb + ((size_t)(kb * 4 + t4) * NKS + KSS * s) * 128 : Qb + ((size_t)t4 * NKS + KSS * s) * 128) + off;
            dma16a(src, lane16, slab + stage * STAGE_REC + half * HALF_REC + d * 64);
        }
    };

    f32x16 acc_o[MT_W][NT_W];
#pragma unroll
    for (int m = 0; m < MT_W; ++m)
#pragma unroll
        for (int n = 0; n < NT_W; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[m][n][r] = 0.0f;
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};   // for queries (2*qh + j)*32 + l31

    // key-range split: each part keeps its own running (max, sum) and an un-normalised output; k_attn_combine merges them
    const int kb_lo = (int)((long long)nkb * split / nsplit), kb_hi = (int)((long long)nkb * (split + 1) / nsplit);
    int base = 0;                                  // stage of slab 0 of the current key block; slab s sits in stage (base + s) & 1
    issue_S(kb_lo, 0, 0);
    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        // ------------------------------------------------ scores: St tiles (kt_w, 2*qh + j), all channels
        f32x16 st[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[j][r] = 0.0f;
        bf16x8 fa[2][2], fb[2][2][2];              // [set][hl], [set][j][hl]
#pragma unroll 1
        for (int s = 0; s < NSS; ++s) {
            const int stage = (base + s) & 1;
            __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): this wave's pieces of slab s have landed
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // the other stage is free now (slab s-1 / the previous block's P are consumed by every wave): next slab goes out.
            // Staggered between the two waves of a SIMD (w, w + 4): waves 0-3 issue their 8 pieces right behind the barrier, waves 4-7
            // after their first k-step's MFMAs -- one of the pair always feeds the matrix pipe.
            if (s + 1 < NSS && wave < 4) issue_S(kb, s + 1, stage ^ 1);
            const u32x4* ka = slab + stage * STAGE_REC + (kt_w * KSS * 2) * 64 + lane;
            const u32x4* qa = slab + stage * STAGE_REC + HALF_REC + (2 * qh * KSS * 2) * 64 + lane;
            auto load_ks = [&](int set, int ks) {
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    fa[set][hl] = __builtin_bit_cast(bf16x8, ka[(ks * 2 + hl) * 64]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[set][j][hl] = __builtin_bit_cast(bf16x8, qa[((j * KSS + ks) * 2 + hl) * 64]);
                }
            };
            load_ks(0, 0);
#pragma unroll
            for (int ks = 0; ks < KSS; ++ks) {
                const int set = ks & 1;
                MDT_PIN();
                if (ks + 1 < KSS) load_ks(set ^ 1, ks + 1);
                MDT_PIN();
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        st[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][term == 0 ? 1 : 0], fb[set][j][term == 1 ? 1 : 0], st[j], 0, 0, 0);
                MDT_PIN();
                if (ks == 0 && s + 1 < NSS && wave >= 4) issue_S(kb, s + 1, stage ^ 1);
            }
        }
        // ------------------------------------------------ V^T fragments of the first two 16-key steps go out now (latency under the softmax)
        const u32x4* vsrc = Vb + ((size_t)kb * 8 * NMT + wm * MT_W) * 128 + lane;      // + p * NMT * 128 + (m * 2 + hl) * 64
        u32x4 fv[3][MT_W][2];
        auto load_v = [&](int set, int p) {
#pragma unroll
            for (int m = 0; m < MT_W; ++m)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) fv[set][m][hl] = vsrc[(size_t)p * NMT * 128 + (m * 2 + hl) * 64];
        };
        load_v(0, 0);
        load_v(1, 1);
        // ------------------------------------------------ online softmax (lane = query column, 16 keys of tile kt_w per j)
        // (scores are kept in the log2 domain: s' = s * scale * log2(e), so that exp(s - m) = exp2(s' - m') is one v_exp_f32)
        const int key0 = kb * BK + kt_w * 32 + 4 * kg;
        const float scale2 = scale * 1.4426950408889634f;
        const bool ragged = (kb + 1) * BK > Tk;          // only the last key block holds padded keys
        float mx[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sv = st[j][r] * scale2;
                if (ragged && key0 + (r & 3) + 8 * (r >> 2) >= Tk) sv = -INFINITY;
                st[j][r] = sv;
                m = fmaxf(m, sv);
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            mx[j] = m;
            if (kg == 0) smax[kt_w * BQ + (2 * qh + j) * 32 + l31] = m;
        }
        // raw barrier + lgkmcnt(0) only: __syncthreads() would also drain vmcnt, i.e. the V^T prefetch just issued (and, further
        // down, the next slab's DMA).  (Also: every wave is done with the last slab.)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the stage of the last slab is free: the next key block's slab 0 goes there and lands under the softmax + output phase
        if (kb + 1 < kb_hi) issue_S(kb + 1, 0, base ^ 1);
        u32x4* const p_l = slab + base * STAGE_REC;  // P records of this key block: the stage slab NSS-2 sat in
        float alpha[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = (2 * qh + j) * 32 + l31;
            const float m_blk = fmaxf(fmaxf(smax[q], smax[BQ + q]), fmaxf(smax[2 * BQ + q], smax[3 * BQ + q]));
            const float m_new = fmaxf(m_run[j], m_blk);      // finite: every key block holds >= 1 valid key
            alpha[j] = __builtin_amdgcn_exp2f(m_run[j] - m_new);   // exp2(-inf) = 0 on the first block
            m_run[j] = m_new;
            float ps = 0.0f;
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pv[r] = __builtin_amdgcn_exp2f(st[j][r] - m_new);   // masked keys: exp2(-inf) = 0
                ps += pv[r];
            }
            ps += __shfl_xor(ps, 32, 64);
            if (kg == 0) {
                ssum[kt_w * BQ + q] = ps;
                if (kt_w == 0) salpha[q] = alpha[j];
            }
            // P records for the output MFMAs: k-step (kt_w*2 + s2) of this key block, query tile 2*qh + j.
            // register r = 8*s2 + i  <->  key 16*s2 + (i & 3) + 8*(i >> 2) + 4*kg of the tile == the Vrec key order.
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v8[i] = pv[8 * s2 + i];
                u32x4 hi, lo;
                split8v(v8, hi, lo);
                u32x4* d = p_l + (((kt_w * 2 + s2) * 4 + (2 * qh + j)) * 2) * 64;
                d[lane] = hi;
                d[64 + lane] = lo;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): P records and the partial sums are written
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = (2 * qh + j) * 32 + l31;
            l_run[j] = l_run[j] * alpha[j] + ((ssum[q] + ssum[BQ + q]) + (ssum[2 * BQ + q] + ssum[3 * BQ + q]));
        }
        // ------------------------------------------------ output: rescale, then Ot += V^T P^T over the 128 keys (no barrier inside)
        {
            // the running maximum of a query settles after a few key blocks: when no query of this wave's tiles moved (alpha == 1
            // everywhere) the MT_W * NT_W * 16 multiplies are skipped (wave-uniform branch)
            float an[NT_W];
            bool moved = false;
#pragma unroll
            for (int n = 0; n < NT_W; ++n) {
                an[n] = salpha[(wn * NT_W + n) * 32 + l31];
                moved = moved || an[n] != 1.0f;
            }
            if (__any(moved)) {
#pragma unroll
                for (int n = 0; n < NT_W; ++n)
#pragma unroll
                    for (int m = 0; m < MT_W; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc_o[m][n][r] *= an[n];
            }
        }
        constexpr int HN = NT_W >= 2 ? NT_W / 2 : 1, NH = NT_W / HN;     // query tiles per half-step, half-steps per 16-key step
        bf16x8 fp[2][HN][2];                                              // [set][n][hl]
        auto load_p = [&](int set, int p, int h) {
#pragma unroll
            for (int n = 0; n < HN; ++n)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    fp[set][n][hl] = __builtin_bit_cast(bf16x8, p_l[((p * 4 + wn * NT_W + h * HN + n) * 2 + hl) * 64 + lane]);
        };
        load_p(0, 0, 0);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int vs = p % 3;
            if (p + 2 < 8) load_v((p + 2) % 3, p + 2);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int t = p * NH + h, ps_ = t & 1;
                MDT_PIN();
                if (t + 1 < 8 * NH) load_p(ps_ ^ 1, (t + 1) / NH, (t + 1) % NH);
                MDT_PIN();
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int n = 0; n < HN; ++n)
#pragma unroll
                        for (int m = 0; m < MT_W; ++m)
                            acc_o[m][h * HN + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, fv[vs][m][term == 0 ? 1 : 0]), fp[ps_][n][term == 1 ? 1 : 0], acc_o[m][h * HN + n], 0, 0, 0);
                MDT_PIN();
            }
        }
        base ^= 1;
    }

    // ---------------------------------------------------- normalise by the softmax denominator and store [B, C, T]
    __syncthreads();
    if (kt_w == 0 && kg == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ql = (2 * qh + j) * 32 + l31;
            ssum[ql] = l_run[j];
            if (nsplit > 1) {   // (max, sum) of this part, per query: pstat[split][b][2][T128]
                float* ps = pstat + ((size_t)(split * gridDim.y + b) * 2) * T128 + qb * BQ + ql;
                ps[0] = m_run[j];
                ps[T128] = l_run[j];
            }
        }
    }
    __syncthreads();
    if (nsplit > 1) {
        float* ob = part + (size_t)(split * gridDim.y + b) * C * T;   // un-normalised part, same [C][T] layout as out
#pragma unroll
        for (int n = 0; n < NT_W; ++n) {
            const int q = qb * BQ + (wn * NT_W + n) * 32 + l31;
            if (q < T) {
#pragma unroll
                for (int m = 0; m < MT_W; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = (wm * MT_W + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                        ob[(size_t)c * T + q] = acc_o[m][n][r];
                    }
            }
        }
        return;
    }
    float* ob = out + (size_t)b * C * T;
#pragma unroll
    for (int n = 0; n < NT_W; ++n) {
        const int ql = (wn * NT_W + n) * 32 + l31;
        const int q = qb * BQ + ql;
        const float inv = 1.0f / ssum[ql];
        if (q < T) {
#pragma unroll
            for (int m = 0; m < MT_W; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = (wm * MT_W + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    ob[(size_t)c * T + q] = acc_o[m][n][r] * inv;
                }
        }
    }
}

// merge the key-range parts of a query: out = sum_s O_s e^(m_s - m) / sum_s l_s e^(m_s - m), m = max_s m_s
__global__ __launch_bounds__(256) void k_attn_combine(const float* __restrict__ part, const float* __restrict__ pstat, float* __restrict__ out,
                                                      int B, int C, int T, int T128, int nsplit) {
    const int q = blockIdx.x * 256 + threadIdx.x, b = blockIdx.z;
    if (q >= T) return;
    float w[4], m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, pstat[((size_t)(s * B + b) * 2) * T128 + q]);
    float den = 0.0f;
    for (int s = 0; s < nsplit; ++s) {
        const float* ps = pstat + ((size_t)(s * B + b) * 2) * T128 + q;
        w[s] = exp2f(ps[0] - m);                     // the parts' maxima are kept in the log2 domain (see the softmax above)
        den += ps[T128] * w[s];
    }
    const float inv = 1.0f / den;
    for (int c = blockIdx.y; c < C; c += gridDim.y) {
        float o = 0.0f;
        for (int s = 0; s < nsplit; ++s) o += part[((size_t)(s * B + b) * C + c) * T + q] * w[s];
        out[((size_t)b * C + c) * T + q] = o * inv;
    }
}

}  // namespace

namespace mdt {

bool attn_bf16x3_eligible(int C) { return C == 128 || C == 256 || C == 512; }

// Key-range split factor: 1 block per CU (133 KB LDS), so the launch runs in ceil(blocks / CUs) rounds; pick the smallest
// nsplit <= 4 whose round occupancy is within 3 % of the best.  MDTILE_ATTN_SPLIT=n forces it.
static int attn_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        return cus;
    }();
    return n;
}
static int attn_nsplit(int B, int Tq, int Tk) {
    static const int forced = [] { const char* e = probe_env("MDTILE_ATTN_SPLIT"); return e ? atoi(e) : 0; }();
    const int nkb = (Tk + 127) / 128;
    if (forced >= 1 && forced <= 4) return forced <= nkb ? forced : 1;
    // The split is chosen for ONE image of the batch, whatever B is: the key ranges fix the order in which a query's partial sums are
    // combined, and a tile's result must not depend on which other tiles it is stacked with (scripts/tilevae.py stacks tiles of one
    // shape along the batch axis; the live-window sweep and the whole-tile sweep stack differently and are compared bit for bit).
    // More images only add blocks to a launch whose split already fills the chip.
    (void)B;
    const long long blocks = (Tq + 127) / 128;
    // launches that fill the chip several times over: 4 key ranges per query block keep the Q working set of an XCD
    // (32 / nsplit query blocks x C x 512 B) inside its L2 -- see the block mapping in k_attn_bf16x3
    if (blocks >= 2 * attn_num_cus() && nkb >= 32) return 4;
    const int cus = attn_num_cus();
    double eff[5], best = 0.0;
    for (int s = 1; s <= 4; ++s) {
        const long long rounds = (blocks * s + cus - 1) / cus;
        eff[s] = s <= nkb ? (double)(blocks * s) / (double)(rounds * cus) : 0.0;
        if (eff[s] > best) best = eff[s];
    }
    for (int s = 1; s <= 4; ++s)
        if (eff[s] >= best - 0.03) return s;
    return 1;
}

// workspace: Qrec (B * Tq128 * C * 4 bytes: hi + lo bf16 per element), Krec, Vrec (B * Tk128 * C * 4 bytes each)
// [+ nsplit un-normalised parts B*C*Tq fp32 and their (max, sum) rows 2*B*Tq128 fp32 when the key range is split]
size_t attn_bf16x3_ws_bytes(int B, int C, int Tq, int Tk) {
    const size_t Tq128 = ((size_t)Tq + 127) / 128 * 128, Tk128 = ((size_t)Tk + 127) / 128 * 128;
    const int ns = attn_nsplit(B, Tq, Tk);
    size_t bytes = (size_t)B * (Tq128 + 2 * Tk128) * C * 4;
    if (ns > 1) bytes += (size_t)ns * ((size_t)B * C * Tq + 2 * (size_t)B * Tq128) * 4;
    return bytes;
}

int attn_bf16x3_launch(const float* d_q, const float* d_k, const float* d_v_tok, float* d_out, int B, int C, int Tq, int Tk, float scale,
                       void* d_ws, hipStream_t s, bool v_channel_major) {
    const int Tq128 = (Tq + 127) / 128 * 128, Tk128 = (Tk + 127) / 128 * 128;
    const size_t perq = (size_t)B * Tq128 * C * 4 / 16, perk = (size_t)B * Tk128 * C * 4 / 16;   // records per operand
    u32x4* Qr = (u32x4*)d_ws;
    u32x4* Kr = Qr + perq;
    u32x4* Vr = Kr + perk;
    const int ns = attn_nsplit(B, Tq, Tk);
    float* part = (float*)(Vr + perk);
    float* pstat = part + (size_t)ns * B * C * Tq;
    {
        const int qtiles = Tq128 / 32, ktiles = Tk128 / 32;
        const size_t nq = (size_t)qtiles * (C / 16) * 64, nk = (size_t)ktiles * (C / 16) * 64;
        hipLaunchKernelGGL(k_attn_prep_qk, dim3(cdiv((long long)nq, 256), B), dim3(256), 0, s, d_q, Qr, C, Tq, qtiles);
        hipLaunchKernelGGL(k_attn_prep_qk, dim3(cdiv((long long)nk, 256), B), dim3(256), 0, s, d_k, Kr, C, Tk, ktiles);
    }
    {
        const int groups = Tk128 / 16;
        const size_t n = (size_t)groups * (C / 32) * 64;
        dim3 grid(cdiv((long long)n, 256), B);
        if (v_channel_major) hipLaunchKernelGGL(k_attn_prep_v_cm, grid, dim3(256), 0, s, d_v_tok, Vr, C, Tk, groups);
        else hipLaunchKernelGGL(k_attn_prep_v, grid, dim3(256), 0, s, d_v_tok, Vr, C, Tk, groups);
    }
    MDT_LAUNCH_CHECK();
    const int nq8 = (Tq128 / BQ + 7) / 8 * 8;
    dim3 grid(nq8 * ns, B), block(512);
#define MDT_ATTN_LAUNCH(CC) hipLaunchKernelGGL((k_attn_bf16x3<CC>), grid, block, 0, s, Qr, Kr, Vr, d_out, Tq, Tq128, Tk, Tk128, scale, ns, part, pstat)
    if (C == 512) MDT_ATTN_LAUNCH(512);
    else if (C == 256) MDT_ATTN_LAUNCH(256);
    else MDT_ATTN_LAUNCH(128);
#undef MDT_ATTN_LAUNCH
    MDT_LAUNCH_CHECK();
    if (ns > 1) {
        dim3 cgrid(cdiv(Tq, 256), C < 64 ? C : 64, B);
        hipLaunchKernelGGL(k_attn_combine, cgrid, dim3(256), 0, s, part, pstat, d_out, B, C, Tq, Tq128, ns);
        MDT_LAUNCH_CHECK();
    }
    return MDTILE_OK;
}

}  // namespace mdt
