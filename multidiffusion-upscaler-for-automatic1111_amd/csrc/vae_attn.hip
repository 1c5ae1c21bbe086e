// Per-tile single-head self-attention of the VAE mid block (K14) as ONE flash-style kernel on the fp32 matrix cores.
// Upstream (tile_utils/attn.py:49-72 and its five host-dependent variants :74-183) materialises the T x T score matrix
// per tile (23.9 GB fp32 for a 278x278-token tile); here it never exists: per 64-query block the kernel streams
// 64-key blocks, keeps running max / sum per query (online softmax) and the 64 x C output accumulator in registers.
//
//   q, k : [B, C, T]  channel-major  (what the 1x1 q/k convs produce in NCHW)
//   v    : [B, T, C]  token-major    (the v conv writes this layout directly, mdtile_conv2d out_layout = 1)
//   out  : [B, C, T]
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact fp32):
//   scores, computed TRANSPOSED:  St[j][i] = sum_c k[c][j] * q[c][i]    A = k (M = keys), B = q (N = queries), K = channels.
//       Both operands are channel-major in LDS -> conflict-free 32-lane ds_read_b32 runs.  In the accumulator a lane then
//       holds ONE query (col = lane & 31) and 16 keys, so softmax row statistics are in-lane + one 32-lane swap.
//   output:  Ot[c][i] += sum_j v[j][c] * P[i][j]                        A = v^T (M = channels), B = P^T (N = queries), K = keys.
//       P^T[j][i] is exactly the score-accumulator layout, so P goes to LDS with plain conflict-free stores, and the
//       result rows are channels / columns are tokens = coalesced 128-B stores into [B, C, T].
// Work split: 4 waves; for the scores each wave owns one 32x32 tile of the 64x64 block; for the output each wave owns
// C/4 channels x 64 queries (C = 512: 8 accumulator tiles = 128 AGPRs).
#include "common.h"

using namespace mdt;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 64;    // queries per block
constexpr int BNK = 64;   // keys per iteration
constexpr int KCQ = 32;   // channels per score-phase slab
constexpr int VJ = 16;    // keys per value-phase slab

template <int CT>  // C = 128 * CT
__global__ __launch_bounds__(256) void k_attn(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                              float* __restrict__ out, int T, float scale) {
    constexpr int C = 128 * CT;
    constexpr int MT = CT;  // 32-channel output tiles per wave (C/4/32)
    // LDS carve (floats)
    constexpr int QS = 0;                          // [2][KCQ][BM]
    constexpr int KS = QS + 2 * KCQ * BM;          // [2][KCQ][BNK]
    constexpr int PT = KS + 2 * KCQ * BNK;         // [BNK][BM]   P^T
    constexpr int VS = PT + BNK * BM;              // [2][VJ][C]
    constexpr int SMAX = VS + 2 * VJ * C;          // [2][BM]
    constexpr int SSUM = SMAX + 2 * BM;            // [2][BM]
    constexpr int ALPHA = SSUM + 2 * BM;           // [BM]
    constexpr int TOTAL = ALPHA + BM;
    __shared__ __attribute__((aligned(16))) float smem[TOTAL];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int jt = wave & 1, it = wave >> 1;
    const int b = blockIdx.y;
    const int i0 = blockIdx.x * BM;
    const float* qb = q + (size_t)b * C * T;
    const float* kb = k + (size_t)b * C * T;
    const float* vb = v + (size_t)b * T * C;

    f32x16 acc_o[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[m][n][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;  // for query i0 + it*32 + l31 (replicated in both lane halves and both jt waves)

    // staging maps: score phase, per thread 8 q + 8 k elements of a [KCQ][64] slab (coalesced 256-B rows)
    constexpr int NQ = KCQ * BM / 256;  // 8
    float rq[NQ], rk[NQ];
    auto load_qk = [&](int c0, int j0) {
#pragma unroll
        for (int e = 0; e < NQ; ++e) {
            const int idx = tid + 256 * e, kc = idx >> 6, col = idx & 63;
            const int qi = i0 + col, kj = j0 + col;
            rq[e] = qi < T ? qb[(size_t)(c0 + kc) * T + qi] : 0.0f;
            rk[e] = kj < T ? kb[(size_t)(c0 + kc) * T + kj] : 0.0f;
        }
    };
    auto store_qk = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NQ; ++e) {
            const int idx = tid + 256 * e;
            smem[QS + buf * KCQ * BM + idx] = rq[e];
            smem[KS + buf * KCQ * BNK + idx] = rk[e];
        }
    };
    constexpr int NV = VJ * C / 4 / 256;  // float4 per thread per value slab (C=512: 8)
    float4 rv[NV];
    auto load_v = [&](int j0) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int f = tid + 256 * e, jj = f / (C / 4), c4 = f - jj * (C / 4);
            const int j = j0 + jj;
            rv[e] = j < T ? *reinterpret_cast<const float4*>(vb + (size_t)j * C + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_v = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NV; ++e) reinterpret_cast<float4*>(smem + VS + buf * VJ * C)[tid + 256 * e] = rv[e];
    };

    const int nkv = (T + BNK - 1) / BNK;
    for (int kv = 0; kv < nkv; ++kv) {
        const int j0 = kv * BNK;
        // ---------------- scores: St tile (jt, it) over all channels ----------------
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.0f;
        constexpr int NCH = C / KCQ;
        load_qk(0, j0);
        store_qk(0);
        __syncthreads();
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch) {
            if (ch + 1 < NCH) load_qk((ch + 1) * KCQ, j0);
            const float* qs = smem + QS + (ch & 1) * KCQ * BM + hi * BM + it * 32 + l31;
            const float* ks = smem + KS + (ch & 1) * KCQ * BNK + hi * BNK + jt * 32 + l31;
#pragma unroll
            for (int s = 0; s < KCQ / 2; ++s)
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[2 * s * BNK], qs[2 * s * BM], st, 0, 0, 0);
            if (ch + 1 < NCH) store_qk((ch + 1) & 1);
            __syncthreads();
        }
        // ---------------- online softmax (lane = one query, 16 of the tile's 32 keys) ----------------
        float sv[16];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            sv[r] = j < T ? st[r] * scale : -INFINITY;
            mx = fmaxf(mx, sv[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (hi == 0) smem[SMAX + jt * BM + it * 32 + l31] = mx;
        // value slab 0 can fly while the statistics are exchanged
        load_v(j0);
        __syncthreads();
        const float m_blk = fmaxf(smem[SMAX + it * 32 + l31], smem[SMAX + BM + it * 32 + l31]);
        const float m_new = fmaxf(m_run, m_blk);       // finite: every key block holds >= 1 valid key
        const float alpha = expf(m_run - m_new);       // exp(-inf) = 0 on the first block
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(sv[r] - m_new);       // masked keys: exp(-inf) = 0
            ps += p;
            smem[PT + (jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * BM + it * 32 + l31] = p;
        }
        ps += __shfl_xor(ps, 32, 64);
        if (hi == 0) {
            smem[SSUM + jt * BM + it * 32 + l31] = ps;
            if (jt == 0) smem[ALPHA + it * 32 + l31] = alpha;
        }
        store_v(0);
        __syncthreads();
        l_run = l_run * alpha + (smem[SSUM + it * 32 + l31] + smem[SSUM + BM + it * 32 + l31]);
        m_run = m_new;
        // ---------------- output: rescale, then Ot += v^T * P^T over the 64 keys ----------------
        {
            const float a0 = smem[ALPHA + l31], a1 = smem[ALPHA + 32 + l31];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc_o[m][0][r] *= a0;
                    acc_o[m][1][r] *= a1;
                }
        }
        constexpr int NVS = BNK / VJ;
#pragma unroll 1
        for (int vs_i = 0; vs_i < NVS; ++vs_i) {
            if (vs_i + 1 < NVS) load_v(j0 + (vs_i + 1) * VJ);
            const float* vs = smem + VS + (vs_i & 1) * VJ * C + hi * C + wave * (C / 4) + l31;
            const float* pt = smem + PT + (vs_i * VJ + hi) * BM + l31;
#pragma unroll
            for (int s = 0; s < VJ / 2; ++s) {
                float av[MT], bp[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = vs[2 * s * C + m * 32];
                bp[0] = pt[2 * s * BM];
                bp[1] = pt[2 * s * BM + 32];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    acc_o[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bp[0], acc_o[m][0], 0, 0, 0);
                    acc_o[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bp[1], acc_o[m][1], 0, 0, 0);
                }
            }
            if (vs_i + 1 < NVS) store_v((vs_i + 1) & 1);
            __syncthreads();
        }
    }

    // ---------------- normalise by the softmax denominator and store [B, C, T] ----------------
    if (jt == 0 && hi == 0) smem[SSUM + it * 32 + l31] = l_run;
    __syncthreads();
    const float inv0 = 1.0f / smem[SSUM + l31], inv1 = 1.0f / smem[SSUM + 32 + l31];
    float* ob = out + (size_t)b * C * T;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = wave * (C / 4) + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int ia = i0 + l31, ib = i0 + 32 + l31;
            if (ia < T) ob[(size_t)c * T + ia] = acc_o[m][0][r] * inv0;
            if (ib < T) ob[(size_t)c * T + ib] = acc_o[m][1][r] * inv1;
        }
}

}  // namespace

// vae_attn_bf16x3.hip
namespace mdt {
bool attn_bf16x3_eligible(int C);
size_t attn_bf16x3_ws_bytes(int B, int C, int Tq, int Tk);
int attn_bf16x3_launch(const float* d_q, const float* d_k, const float* d_v_tok, float* d_out, int B, int C, int Tq, int Tk, float scale,
                       void* d_ws, hipStream_t s, bool v_channel_major = false);
}  // namespace mdt

static bool attn_force_f32() { return attn_strict_f32(); }

// the flash formulation keeps every intermediate on chip; the split-bf16 path needs room for the fragment-order
// bf16 hi/lo images of q, k and v
extern "C" size_t mdtile_vae_attn_ws_size(int B, int C, int T) {
    if (B <= 0 || T <= 0 || !attn_bf16x3_eligible(C)) return 0;
    return attn_bf16x3_ws_bytes(B, C, T, T);
}

// one predicate for the dispatch below AND for the host's choice of the v layout (round 3's host-side test looked at mdtile_get_precision,
// which reports F32 only when conv AND attention are forced: MDTILE_ATTN_MODE=f32 alone sent a channel-major v to the exact kernel)
static bool attn_takes_bf16x3(int C, int flags) { return !(flags & MDTILE_ATTN_EXACT_F32) && !attn_force_f32() && attn_bf16x3_eligible(C); }

extern "C" int mdtile_vae_attn_takes_channel_major(int C, int flags) { return attn_takes_bf16x3(C, flags) ? 1 : 0; }

extern "C" int mdtile_vae_attn(const float* d_q, const float* d_k, const float* d_v, float* d_out, int B, int C, int T, float scale,
                               int flags, void* d_ws, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_q && d_k && d_v && d_out, "mdtile_vae_attn: null argument");
    MDT_CHECK_ARG(B > 0 && B <= 65535 && T > 0, "mdtile_vae_attn: bad shape B=%d T=%d", B, T);
    MDT_CHECK_ARG(C == 128 || C == 256 || C == 512, "mdtile_vae_attn: C=%d unsupported (128, 256 or 512)", C);
    hipStream_t s = as_stream(stream);
    if (attn_takes_bf16x3(C, flags)) {
        MDT_CHECK_ARG(d_ws, "mdtile_vae_attn: the split-bf16 path needs the workspace of mdtile_vae_attn_ws_size()");
        return attn_bf16x3_launch(d_q, d_k, d_v, d_out, B, C, T, T, scale, d_ws, s, (flags & MDTILE_ATTN_V_CHANNEL_MAJOR) != 0);
    }
    MDT_CHECK_ARG(!(flags & MDTILE_ATTN_V_CHANNEL_MAJOR), "mdtile_vae_attn: the exact-fp32 kernel takes v token-major");
    dim3 grid((T + BM - 1) / BM, B), block(256);
    if (C == 512) hipLaunchKernelGGL(k_attn<4>, grid, block, 0, s, d_q, d_k, d_v, d_out, T, scale);
    else if (C == 256) hipLaunchKernelGGL(k_attn<2>, grid, block, 0, s, d_q, d_k, d_v, d_out, T, scale);
    else hipLaunchKernelGGL(k_attn<1>, grid, block, 0, s, d_q, d_k, d_v, d_out, T, scale);
    MDT_LAUNCH_CHECK();
    return MDTILE_OK;
}

// Queries of one row band against keys / values of the whole image (q [B,C,Tq]; k [B,C,Tk]; v [B,Tk,C]; out [B,C,Tq]):
// the attention step of the sequence-parallel fast-mode estimator (mdtile/seqpar.py).  Split-bf16 kernel only.
extern "C" size_t mdtile_vae_attn_qk_ws_size(int B, int C, int Tq, int Tk) {
    if (B <= 0 || Tq <= 0 || Tk <= 0 || !attn_bf16x3_eligible(C)) return 0;
    return attn_bf16x3_ws_bytes(B, C, Tq, Tk);
}

extern "C" int mdtile_vae_attn_qk(const float* d_q, const float* d_k, const float* d_v, float* d_out, int B, int C, int Tq, int Tk,
                                  float scale, void* d_ws, mdtile_stream_t stream) {
    MDT_CHECK_ARG(d_q && d_k && d_v && d_out && d_ws, "mdtile_vae_attn_qk: null argument");
    MDT_CHECK_ARG(B > 0 && B <= 65535 && Tq > 0 && Tk > 0, "mdtile_vae_attn_qk: bad shape B=%d Tq=%d Tk=%d", B, Tq, Tk);
    MDT_CHECK_ARG(attn_bf16x3_eligible(C), "mdtile_vae_attn_qk: C=%d unsupported (128, 256 or 512)", C);
    return attn_bf16x3_launch(d_q, d_k, d_v, d_out, B, C, Tq, Tk, scale, d_ws, as_stream(stream));
}
