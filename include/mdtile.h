/*
 * mdtile.h -- C ABI of the MI355X-native tiled-diffusion / tiled-VAE engine (libmdtile.so).
 *
 * This is the drop-in boundary below the A1111 plugin surface (SURVEY.md section 8b "inner boundary").
 * The upstream extension is pure Python and has no FFI of its own; each entry point below replaces the torch
 * op sequence at the cited upstream call site (paths relative to the upstream repo root).  The Python host
 * (multidiffusion-upscaler-for-automatic1111_amd/mdtile/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain C types only; every device buffer is owned by the caller (PyTorch), the library allocates nothing on
 *     the device except the small index tables inside an opaque plan
 *   - every launch is asynchronous on the hipStream_t passed as `stream` (void* here; NULL = default stream)
 *   - return value: 0 = ok, negative = error (see MDTILE_E_*); mdtile_last_error() gives text; nothing throws/aborts
 *   - tensors are dense NCHW; "f32"/"f16"/"bf16" I/O selected by MDTILE_DT_*; all accumulation is fp32
 *   - not re-entrant on one plan; one plan per host thread
 */
#ifndef MDTILE_H
#define MDTILE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDTILE_VERSION 100

#define MDTILE_OK 0
#define MDTILE_E_ARG (-1)      /* bad argument / unsupported shape */
#define MDTILE_E_HIP (-2)      /* HIP runtime error               */
#define MDTILE_E_LIMIT (-3)    /* compile-time capacity exceeded  */

#define MDTILE_DT_F32 0
#define MDTILE_DT_F16 1
#define MDTILE_DT_BF16 2

#define MDTILE_METHOD_MD 0     /* MultiDiffusion            (tile_methods/multidiffusion.py)      */
#define MDTILE_METHOD_MOD 1    /* Mixture of Diffusers      (tile_methods/mixtureofdiffusers.py)  */

#define MDTILE_REGION_BG 0     /* BlendMode.BACKGROUND (tile_utils/utils.py:36-39) */
#define MDTILE_REGION_FG 1     /* BlendMode.FOREGROUND */

#define MDTILE_MAX_BATCHES 320 /* tile-batch pointers carried in kernel arguments */
#define MDTILE_MAX_REGIONS 16  /* = upstream md_max_regions cap (scripts/tilediffusion.py:81) */

typedef void* mdtile_stream_t; /* hipStream_t */
typedef struct mdtile_plan mdtile_plan;

int mdtile_version(void);
const char* mdtile_last_error(void);

/* Arithmetic of the matrix-core kernels (convs, attention).  Default MDTILE_PRECISION_BF16X3: every fp32 factor is split into two
 * bf16 halves (16 significand bits), three bf16 MFMAs per product, fp32 accumulation (~1e-5 relative to fp32 end to end; the stated
 * tolerance of the path is 1e-3).  MDTILE_PRECISION_F32: exact-fp32 MFMA kernels everywhere (bit-comparable to an fp32 fmaf chain),
 * ~4x slower.  Process-wide; env MDTILE_CONV_MODE=f32 / MDTILE_ATTN_MODE=f32 preset it. */
#define MDTILE_PRECISION_BF16X3 0
#define MDTILE_PRECISION_F32 1
int mdtile_set_precision(int mode);
int mdtile_get_precision(void);

/* ----------------------------------------------------------------------------------------------------------
 * Grid planning (host integers only).
 * Replaces split_bboxes (tile_utils/utils.py:160-177) + init_grid_bbox (tile_methods/abstractdiffusion.py:173-186):
 *   tile := min(tile, canvas); overlap := max(0, min(overlap, min(tile_w_req, tile_h_req) - 4));
 *   cols = ceil((w-ov)/(tw-ov)); dx = (w-tw)/(cols-1) (double); x_c = min(int(c*dx), w-tw); rows likewise;
 *   num_batches = ceil(T/tile_bs); tile_bs := ceil(T/num_batches).
 * w,h,tile_*,overlap are in latent pixels.  The plan owns small device tables (origins, per-column / per-row
 * covering ranges) used by the gather-formulated kernels below.
 * -------------------------------------------------------------------------------------------------------- */
mdtile_plan* mdtile_plan_create(int w, int h, int tile_w, int tile_h, int overlap, int tile_bs, int clamp);
/* clamp = 1: init_grid_bbox semantics (tile and overlap clamped as above); clamp = 0: raw split_bboxes(w,h,tw,th,overlap) */
void mdtile_plan_destroy(mdtile_plan* plan);
/* info[8] = { cols, rows, num_tiles, num_batches, tile_bs, tile_w, tile_h, overlap_effective } */
int mdtile_plan_info(const mdtile_plan* plan, int* info8);
/* xywh[4*num_tiles], row-major tile order (y outer) == upstream bbox list order */
int mdtile_plan_bboxes(const mdtile_plan* plan, int* xywh);

/* ----------------------------------------------------------------------------------------------------------
 * Weight maps (init time, device).
 * -------------------------------------------------------------------------------------------------------- */
/* gaussian_weights (tile_utils/utils.py:180-194): d_out[tile_h*tile_w] fp32; profile evaluated in fp64, var=0.01,
 * both axes normalised by tile_w^2, x midpoint (tile_w-1)/2, y midpoint tile_h/2. */
int mdtile_gaussian_weights(int tile_w, int tile_h, float* d_out, mdtile_stream_t stream);
/* feather_mask (tile_utils/utils.py:196-214): d_out[h*w] fp32. */
int mdtile_feather_mask(int w, int h, double ratio, float* d_out, mdtile_stream_t stream);
/* Grid weight map: the `weight[bbox.slicer] += init_weight` loop of split_bboxes (utils.py:164-175) followed by
 * `self.weights += weights` (abstractdiffusion.py:181-182).  d_tile_w == NULL -> uniform 1.0 (MultiDiffusion),
 * else the [tile_h, tile_w] Gaussian (Mixture of Diffusers).  d_weights[h*w] is updated in place. */
int mdtile_weight_map_add_grid(const mdtile_plan* plan, const float* d_tile_w, float* d_weights, mdtile_stream_t stream);
/* Region contribution: `self.weights[bbox.slicer] += 1.0` (multidiffusion.py:44-46, d_rect_w == NULL, scalar used) or
 * `+= gaussian_weights(bbox.w, bbox.h)` (mixtureofdiffusers.py:50-53).  Canvas is [H, W]. */
int mdtile_weight_map_add_rect(float* d_weights, int W, int H, int x, int y, int w, int h, const float* d_rect_w,
                               float scalar, mdtile_stream_t stream);
/* rescale_factor = 1 / weights (mixtureofdiffusers.py:32); inf where weights == 0, as upstream. */
int mdtile_reciprocal(const float* d_in, float* d_out, size_t n, mdtile_stream_t stream);
/* custom_weights[i] *= rescale_factor[bbox.slicer] (mixtureofdiffusers.py:34-36); d_rect_w[h*w] in place. */
int mdtile_rect_mul_canvas(float* d_rect_w, const float* d_canvas, int W, int H, int x, int y, int w, int h,
                           mdtile_stream_t stream);

/* ----------------------------------------------------------------------------------------------------------
 * Tile gather (K2): x_tile[i*N + n, c, :, :] = x_in[n, c, y_i:y_i+th, x_i:x_i+tw]  (tile-major batch layout)
 * Replaces `torch.cat([x_in[bbox.slicer] for bbox in bboxes], dim=0)` (multidiffusion.py:155,
 * mixtureofdiffusers.py:88,104).  mdtile_gather_all fills every batch in ONE launch (batch_ptrs[num_batches]).
 * -------------------------------------------------------------------------------------------------------- */
int mdtile_gather(const mdtile_plan* plan, int dtype, int N, int C, const void* d_x_in, int batch_id, void* d_x_tile,
                  mdtile_stream_t stream);
int mdtile_gather_all(const mdtile_plan* plan, int dtype, int N, int C, const void* d_x_in, void* const* batch_ptrs,
                      int num_batches, mdtile_stream_t stream);
/* Tiles [tile_lo, tile_hi) only, into ONE packed buffer laid out as [T*N, C, th, tw] (tile t at row t*N): the multi-GPU
 * path gathers just the rank's band of tile rows. */
int mdtile_gather_range(const mdtile_plan* plan, int dtype, int N, int C, const void* d_x_in, void* d_packed, int tile_lo,
                        int tile_hi, mdtile_stream_t stream);
/* Arbitrary-rect gather for custom regions: `x_in[bbox.slicer]` (multidiffusion.py:184, mixtureofdiffusers.py:140). */
int mdtile_gather_rect(int dtype, int N, int C, int W, int H, const void* d_x_in, int x, int y, int w, int h,
                       void* d_out, mdtile_stream_t stream);
/* Plain streaming copy of `bytes` bytes (16-byte accesses, one KiB per wave-instruction; both pointers 16-byte aligned, bytes % 16 == 0),
 * launched like the blend (one grid over the buffer, same stream).  Nothing upstream corresponds to it: it is the measurement
 * floor bench.py prints beside the blend kernel (`roofline_blend.copy_floor_us`: the same number of HBM bytes moved with no
 * table hop, no tile walk and no arithmetic), and a device-to-device copy for callers that have no torch at hand. */
int mdtile_stream_copy(const void* d_src, void* d_dst, size_t bytes, mdtile_stream_t stream);

/* ----------------------------------------------------------------------------------------------------------
 * Overlap blend (K3..K7), gather-formulated: one thread owns output pixels, sums the covering tile values in
 * upstream's tile order (so fp32 results are bit-identical to the sequential `+=` loop), then applies the
 * method's epilogue.  ONE launch per model evaluation.
 *
 *   MD  (multidiffusion.py:147-216):  buf = sum_tiles out_t (+ sum_bg out_r);  x = weights > 1 ? buf / weights : buf
 *   MoD (mixtureofdiffusers.py:74-175): buf = sum_tiles out_t * (tile_w * rescale[slicer]) (+ sum_bg out_r * cw_r)
 *   both: foreground regions: fbuf += out_r, fmask += feather_r, fcnt += 1; averaged where fcnt > 1;
 *         x = fcnt > 0 ? x * (1 - fmask) + fbuf * fmask : x
 *
 * batch_out[b] : device pointer to the model output of tile batch b, [nb_b*N, C, tile_h, tile_w] (tile-major)
 *                num_batches == 0  <=>  draw_background == False (abstractdiffusion.py:199-201)
 * regions      : in upstream list order
 * flags        : MDTILE_BLEND_PARTIAL = write raw partial sums (buf only, no epilogue) -- the multi-GPU path sums
 *                partials across ranks and then calls mdtile_blend_finalize.
 * -------------------------------------------------------------------------------------------------------- */
typedef struct mdtile_region {
    int x, y, w, h;
    int mode;            /* MDTILE_REGION_BG | MDTILE_REGION_FG */
    int _pad;
    const void* out;     /* model output for the region, [N, C, h, w], dtype as the call */
    const float* weight; /* BG+MoD: pre-multiplied custom weight [h,w]; FG: feather mask [h,w]; BG+MD: NULL */
} mdtile_region;

#define MDTILE_BLEND_PARTIAL 1
#define MDTILE_BLEND_TILE_RANGE 2 /* only tiles with tile_lo <= index < tile_hi contribute (row-band sharding) */
#define MDTILE_BLEND_PACKED 4     /* batch_out[0] is ONE packed buffer [T*N, C, th, tw] holding every tile (any T) */

typedef struct mdtile_blend_args {
    int method;                  /* MDTILE_METHOD_* */
    int dtype;                   /* MDTILE_DT_* of batch_out / region out / x_out */
    int N, C;
    int flags;
    int tile_lo, tile_hi;        /* with MDTILE_BLEND_TILE_RANGE */
    int row_lo, row_hi;          /* canvas rows to produce [row_lo,row_hi); 0,0 = all */
    const float* d_weights;      /* MD : [H,W] weight-sum map                                        */
    const float* d_tile_w;       /* MoD: [tile_h, tile_w] Gaussian                                    */
    const float* d_rescale;      /* MoD: [H,W] 1/weights                                              */
    void* d_x_out;               /* [N,C,H,W]; with PARTIAL: fp32 [N,C,H,W] raw sums                  */
} mdtile_blend_args;

int mdtile_blend(const mdtile_plan* plan, const mdtile_blend_args* args, const void* const* batch_out, int num_batches,
                 const mdtile_region* regions, int num_regions, mdtile_stream_t stream);
/* Epilogue on summed partials (multi-GPU): MD divide + FG composite, identical math to mdtile_blend's tail.
 * d_partial fp32 [N,C,H,W]; FG regions are re-read from `regions`. */
int mdtile_blend_finalize(const mdtile_plan* plan, const mdtile_blend_args* args, const float* d_partial,
                          const mdtile_region* regions, int num_regions, mdtile_stream_t stream);

/* Per-region initial noise (scripts/tilediffusion.py:486-529, create_random_tensors_hijack): d_noise [N,C,H,W] fp32 is the job's
 * noise, updated in place.  regions[i].out = the region's own noise [1,C,h,w] fp32 (drawn by the host with the region's CPU
 * seed, shared by all N samples), .mode = MDTILE_REGION_BG / _FG, .weight unused.  Per layer: sum + hit count over the regions
 * in list order, averaged where count > 1; background layer pasted where its count > 0, foreground layer on top.  <= 16 regions. */
int mdtile_region_noise(float* d_noise, int N, int C, int H, int W, const mdtile_region* regions, int num_regions,
                        mdtile_stream_t stream);

/* Noise Inversion renoise composite (tile_methods/abstractdiffusion.py:651-676, inside sample_img2img):
 *   d_noise, d_inverse_noise, d_out [N,C,H,W] fp32, d_renoise_mask [H,W] fp32 (already scaled by the renoise strength and clamped).
 *   num_regions > 0 (the caller passes the custom regions only when the grid is disabled, :658): the job's noise is first re-weighted
 *   noise' = bg * (1 - fw) + fg * fw with bg = noise where a background region covers the pixel, fg / fw = hit-count averages of
 *   the noise / of the foreground regions' feather masks (.weight [h,w] fp32; .out unused), in list order.
 *   out = ((1 - m) * inverse + m * noise') / sqrt(m^2 + (1 - m)^2).   The same fp32 ops in the same order as the eager code, each
 *   correctly rounded (torch's CPU sqrt is 1 ulp off on ~1 % of inputs: that is the only possible last-bit difference).  <= 16 regions. */
int mdtile_noise_inverse_blend(const float* d_noise, const float* d_inverse_noise, const float* d_renoise_mask, float* d_out, int N, int C,
                               int H, int W, const mdtile_region* regions, int num_regions, mdtile_stream_t stream);

/* ControlNet / StableSR tile slicing (tile_methods/abstractdiffusion.py:475-544, 548-588): num_rects (<= 16) rectangles of size w x h
 * at rects_xy[2 i], rects_xy[2 i + 1] of d_x_in [N,C,H,W] are cut out, concatenated (tile-major, then the N samples: torch.cat over
 * the bboxes) and repeated `repeat` times for the sampler's cond / uncond copies:
 *   tile_major = 1: out row j * repeat + r = row j (k-diffusion, :528-533);   tile_major = 0: out row r * (num_rects * N) + j (DDIM, :535).
 * d_out [num_rects * N * repeat, C, h, w], same dtype.  Pass the latent-grid rectangles multiplied by opt_f = 8 for ControlNet hints. */
int mdtile_gather_rects(int dtype, int N, int C, int W, int H, const void* d_x_in, const int* rects_xy, int num_rects, int w, int h,
                        int repeat, int tile_major, void* d_out, mdtile_stream_t stream);

/* DemoFusion model evaluation (tile_methods/demofusion.py:219-324).  Tensors NCHW in `dtype`, fp32 accumulation; Hp x Wp is the latent padded
 * by the jitter range J on every side (forward_one_step, :201-205).
 *   mdtile_window_blend       local path (:244-257): the T = rows*cols equally sized windows (origins d_window_xy[2 w], [2 w + 1] in the padded
 *                             canvas, row-major; nominal, un-jittered origins d_nomx[cols] / d_nomy[rows]; every origin lies within
 *                             [nominal, nominal + 2 J]) are summed in list order and divided by the hit count (0 -> 1).
 *                             d_tiles [T*N, C, window, window] tile-major (the concatenated model outputs), d_out [N,C,Hp,Wp].
 *   mdtile_dilated_gather     global path, inputs (:268-283): cell i = (cells_xy[2 i], cells_xy[2 i + 1]) of the S x S lattice ->
 *                             x[:, :, by+J : Wp-J : S, bx+J : Wp-J : S]; the first num_from_x cells read d_x, the others d_x_filtered (mixture
 *                             mode); d_out [num_cells*N, C, h0, w0], cells in list order.
 *   mdtile_demofusion_combine global path, scatter + mix (:284-322): x_global = scatter of d_global_out [cells*N, C, h0, w0] back onto the lattice
 *                             (mixture: the two copies of a cell are added and halved), out = x_local * (1 - c2) + x_global * c2.
 *   mdtile_depthwise_blur     Gaussian filter (:173-178): depthwise K x K conv (odd K), zero padding, kernel [K*K] fp32 on the device.
 *   mdtile_restandardize      (x - st[0]) / st[1] * st[3] + st[2]  with st = { mean, std, target mean, target std } on the device (:264). */
int mdtile_window_blend(int dtype, const void* d_tiles, void* d_out, const int* d_window_xy, const int* d_nomx, const int* d_nomy, int rows,
                        int cols, int jitter, int window, int N, int C, int Hp, int Wp, mdtile_stream_t stream);
int mdtile_dilated_gather(int dtype, const void* d_x, const void* d_x_filtered, int num_from_x, void* d_out, const int* cells_xy, int num_cells,
                          int N, int C, int Hp, int Wp, int S, int jitter, int h0, int w0, mdtile_stream_t stream);
int mdtile_demofusion_combine(int dtype, const void* d_x_local, const void* d_global_out, void* d_out, int N, int C, int Hp, int Wp, int S,
                              int jitter, int h0, int w0, int mixture, float c2, mdtile_stream_t stream);
int mdtile_depthwise_blur(int dtype, const void* d_x, const float* d_kernel, void* d_out, int planes, int H, int W, int K, mdtile_stream_t stream);
int mdtile_restandardize(int dtype, const void* d_x, const float* d_stats4, void* d_out, size_t n, mdtile_stream_t stream);

/* ----------------------------------------------------------------------------------------------------------
 * Tiled VAE (scripts/tilevae.py).  All tensors fp32 NCHW.
 * -------------------------------------------------------------------------------------------------------- */
/* split_tiles + get_best_tile_size (tilevae.py:390-462).  in_bboxes/out_bboxes: [4*n] as [x1,x2,y1,y2].
 * Returns the tile count (>=1) or a negative error; pass cap = capacity in tiles (call with cap=0 to query). */
int mdtile_vae_split_tiles(int h, int w, int tile_size, int is_decoder, int* in_bboxes, int* out_bboxes, int cap);
/* VAEHook.get_best_tile_size (scripts/tilevae.py:390-403): the size split_tiles shrinks a [lowerbound, upperbound] tile to. */
int mdtile_vae_best_tile_size(int lowerbound, int upperbound);

/* get_var_mean (tilevae.py:207-215): biased var & mean per (sample, group); x [B,C,HW]; out [B*groups] each.
 * Deterministic two-stage reduction (fp64 accumulators, no atomics); d_ws: mdtile_gn_stats_ws_size(B, groups) bytes. */
size_t mdtile_gn_stats_ws_size(int B, int groups);
int mdtile_gn_stats(const float* d_x, int B, int C, int HW, int groups, float* d_mean, float* d_var, void* d_ws,
                    mdtile_stream_t stream);
/* GroupNormParam.summary (tilevae.py:320-335): pixel-weighted pooling of T per-tile stat rows.
 * d_means/d_vars [T, BG]; d_frac[T] = the normalised pixel fractions p_i (tilevae.py:328-331, T host-side floats
 * computed by the caller and copied to the device); out [BG]:  var = sum_i p_i var_i, mean = sum_i p_i mean_i. */
int mdtile_gn_pool(const float* d_means, const float* d_vars, const float* d_frac, int T, int BG, float* d_mean,
                   float* d_var, mdtile_stream_t stream);
/* custom_group_norm (tilevae.py:218-245) fused with the following in-place SiLU (tilevae.py:102-104):
 * y = ((x - mean_g) / sqrt(var_g + eps)) * gamma_c + beta_c ; if silu: y = y * sigmoid(y).  gamma/beta may be NULL.
 * In place (d_y == d_x) allowed. */
int mdtile_gn_apply(const float* d_x, float* d_y, int B, int C, int HW, int groups, const float* d_mean,
                    const float* d_var, const float* d_gamma, const float* d_beta, float eps, int silu,
                    mdtile_stream_t stream);
/* standalone SiLU / residual add (queue tasks 'silu', 'add_res', tilevae.py:102-104, 614-616) */
int mdtile_silu(const float* d_x, float* d_y, size_t n, mdtile_stream_t stream);
int mdtile_tanh(const float* d_x, float* d_y, size_t n, mdtile_stream_t stream);   /* Decoder.tanh_out, scripts/tilevae.py:192-193 */
int mdtile_add(const float* d_a, const float* d_b, float* d_y, size_t n, mdtile_stream_t stream);

/* Conv tiles (queue tasks conv_in/conv1/conv2/nin_shortcut/upsample/conv_out/q/k/v/proj_out, tilevae.py:115-195).
 * stride 1, 'same' zero padding, ksize 1 or 3.  Weights are pre-packed once by mdtile_conv_pack (OIHW -> [tap][cin][cout] fp32,
 * followed by the split-bf16 fragment-order image for the shapes the bf16x3 kernel takes; mdtile_conv_packed_size covers both).
 *   MDTILE_CONV_UPSAMPLE2X : input is read through a nearest 2x upsample (ldm Upsample: interpolate then conv)
 *   d_residual != NULL     : y = conv(x) + bias + residual   ('add_res' fused)
 *   out_layout             : 0 = [B,Cout,H,W];  1 = token-major [B,H*W,Cout] (V operand of mdtile_vae_attn)
 * H, W are the OUTPUT spatial size. */
#define MDTILE_CONV_UPSAMPLE2X 1
#define MDTILE_CONV_EXACT_F32 2   /* force the exact-fp32 MFMA kernel (default: split-bf16 "bf16x3" MFMA where the shape allows,
                                     fp32 accumulate, ~1e-5 relative vs fp32; env MDTILE_CONV_MODE=f32 forces it globally) */
size_t mdtile_conv_packed_size(int cout, int cin, int ksize); /* floats */
int mdtile_conv_pack(const float* d_w_oihw, float* d_w_packed, int cout, int cin, int ksize, mdtile_stream_t stream);
int mdtile_conv2d(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual, float* d_y,
                  int B, int cin, int cout, int H, int W, int ksize, int flags, int out_layout, mdtile_stream_t stream);

/* ldm Downsample (encoder 'downsample' task, tilevae.py:155-171): y = conv3x3_stride2(pad(x, right 1, bottom 1)) + bias,
 * y is [B, cout, (Hin-2)/2+1, (Win-2)/2+1].  d_w_packed: mdtile_conv_pack(ksize 3) of the conv's OIHW weights. */
int mdtile_conv2d_down2(const float* d_x, const float* d_w_packed, const float* d_bias, float* d_y, int B, int cin, int cout,
                        int Hin, int Win, mdtile_stream_t stream);

/* Fused pre-activation: the fixed-statistics GroupNorm + SiLU that precedes conv1 / conv2 of every ResnetBlock
 * (custom_group_norm + inplace_nonlinearity, tilevae.py:218-245, 102-104; queue order pre_norm, silu, conv1 ... :115-137)
 * is applied while the conv stages its input, so the normalised activation never makes a round trip through HBM.
 *   mdtile_gn_coeffs           : d_coef[B][2][C] = { a = gamma / sqrt(var + eps), s = beta - mean * a } (same constants as mdtile_gn_apply)
 *   mdtile_conv2d_gn           : y = conv(silu(a * x + s)) + bias (+ residual); 3x3, stride 1, split-bf16 kernel only
 *   mdtile_conv2d_gn_supported : 1 when such a kernel exists for the shape / flags (else use mdtile_gn_apply + mdtile_conv2d) */
int mdtile_gn_coeffs(const float* d_mean, const float* d_var, const float* d_gamma, const float* d_beta, int B, int C, int groups,
                     float eps, float* d_coef, mdtile_stream_t stream);
int mdtile_conv2d_gn_supported(int cout, int cin, int ksize, int flags, int out_layout);
int mdtile_conv2d_gn(const float* d_x, const float* d_coef, const float* d_w_packed, const float* d_bias, const float* d_residual,
                     float* d_y, int B, int cin, int cout, int H, int W, int ksize, int flags, mdtile_stream_t stream);

/* Slow mode (round 5): the conv whose output is the input of a POOLED GroupNorm leaves that output's statistics itself -- what
 * GroupNormParam.add_tile asks of get_var_mean (tilevae.py:207-215, 300-307) without the pass that re-reads the activation.  The kernels'
 * epilogues write (sum, sum of squares) per block and 4-cout quad (fp32 inside a lane's <= 16 values, fp64 from there on); two small kernels combine them in a fixed order
 * (deterministic, no atomics) into d_mean / d_var [B * groups] (biased variance, as mdtile_gn_stats).  d_y is bit-identical to the call
 * without statistics.  d_ws: mdtile_conv_stats_ws_size(B, cout, H, W, groups) bytes (H, W = OUTPUT size), 16-byte aligned.
 *   mdtile_conv2d_gn_stats(_supported)  : mdtile_conv2d_gn + statistics (128-cout blocks: cout % 128 == 0; (cout / groups) % 4 == 0)
 *   mdtile_conv2d_rec_stats(_supported) : mdtile_conv2d_rec (fp32 output only; MDTILE_CONV_UPSAMPLE2X allowed) + statistics, declared below.
 *                                         Launches of only a few item rounds keep the kernel family mdtile_conv2d_rec would choose and
 *                                         run the statistics pass inside the call (same results contract, one entry point). */
size_t mdtile_conv_stats_ws_size(int B, int cout, int H, int W, int groups);
int mdtile_conv2d_gn_stats_supported(int cout, int cin, int ksize, int flags, int groups);
int mdtile_conv2d_gn_stats(const float* d_x, const float* d_coef, const float* d_w_packed, const float* d_bias, const float* d_residual,
                           float* d_y, int B, int cin, int cout, int H, int W, int ksize, int flags, int groups, float* d_mean, float* d_var,
                           void* d_ws, mdtile_stream_t stream);

/* Record-image conv path (fast mode: every GroupNorm's statistics are frozen before the tiles run, tilevae.py:464-505, 542-563).
 * A "record image" of an activation [B, C, H, W] (C % 32 == 0) is its split-bf16 form in MFMA fragment order with a
 * 1-pixel zero border:  rec[b][hl][C/8][H+2][pitch] x 16 bytes, hl = 0: bf16(x), hl = 1: bf16(x - hi); plane p = 2*kstep + kg
 * holds channels 32*(kstep>>1) + 16*(kstep&1) + 4*kg + (j&3) + 8*(j>>2), j = 0..7  -- 4 bytes per element, like fp32.
 * Rows: pitch = (W + 9 + 7) & ~7 records; pixel x sits at column x + 8, the border records at columns 7 and W + 8, the rest of a row
 * is padding nobody reads -- so that every 32-pixel run the kernels store is four whole 128-byte lines (round 4; the size is
 * opaque to callers: mdtile_rec_size).
 * The PRODUCER applies the following norm's (a, s) + SiLU (custom_group_norm + inplace_nonlinearity, tilevae.py:218-245,
 * 102-104) and splits; the consuming conv stages its input by DMA only.
 *   mdtile_rec_size            : bytes of the record image of [B, C, H, W]
 *   mdtile_rec_from_f32        : rec = split(silu(a x + s)) (d_coef = mdtile_gn_coeffs output [B][2][C]) or split(x) (d_coef NULL)
 *   mdtile_rec_to_f32          : x = hi + lo (inspection / tests)
 *   mdtile_conv2d_rec_supported: 1 when the record kernels take the shape (3x3, cin % 32 == 0, cout % 128 == 0; or cout < 32 -- conv_out:
 *                                fp32 output only, no residual / upsample, d_bias padded with zeros to 32 floats by the caller)
 *   mdtile_conv2d_rec          : y = conv3x3(x_rec) + bias (+ residual), written as fp32 NCHW (d_y, may be NULL) and / or as the
 *                                record image d_y_rec = split(silu(a y + s)) (d_y_coef [B][2][cout]) or split(y) (d_y_coef NULL);
 *                                MDTILE_CONV_UPSAMPLE2X: x_rec is the HALF-size input of the fused nearest-2x upsample conv
 *                                (ldm Upsample, tilevae.py:139-153).  H, W = output size.  d_w_packed: mdtile_conv_pack(ksize 3). */
#define MDTILE_CONV_REC_ONE_BLOCK 4   /* mdtile_conv2d_rec / mdtile_upconv2d_rec_window flags: name the kernel family instead of letting the */
#define MDTILE_CONV_REC_TWO_BLOCKS 8  /* launcher choose per launch (one 8-wave block per CU / two 4-wave blocks per CU; identical results) */
size_t mdtile_rec_size(int B, int C, int H, int W);
int mdtile_rec_from_f32(const float* d_x, const float* d_coef, void* d_rec, int B, int C, int H, int W, mdtile_stream_t stream);
int mdtile_rec_to_f32(const void* d_rec, float* d_x, int B, int C, int H, int W, mdtile_stream_t stream);
int mdtile_conv2d_rec_supported(int cout, int cin, int ksize, int flags);
int mdtile_conv2d_rec(const void* d_x_rec, const float* d_w_packed, const float* d_bias, const float* d_residual, float* d_y,
                      void* d_y_rec, const float* d_y_coef, int B, int cin, int cout, int H, int W, int flags,
                      mdtile_stream_t stream);
int mdtile_conv2d_rec_stats_supported(int cout, int cin, int ksize, int flags, int groups);
int mdtile_conv2d_rec_stats(const void* d_x_rec, const float* d_w_packed, const float* d_bias, const float* d_residual, float* d_y,
                            int B, int cin, int cout, int H, int W, int flags, int groups, float* d_mean, float* d_var, void* d_ws,
                            mdtile_stream_t stream);
/* Live-window narrowing of a decoder tile (fast mode only: frozen GroupNorm statistics make every later layer local).  Upstream decodes the
 * whole padded tile and crop_valid_region (tilevae.py:248-259, applied at :630-632) throws the padding away at the end; a pixel the remaining
 * 3x3 convs cannot carry into the valid region need not be computed at all.  The narrowing happens where the resolution doubles:
 *   mdtile_upconv2d_rec_window : the fused nearest-2x upsample conv (ldm Upsample, tilevae.py:139-153) of the window
 *                                [y0[b] : y0[b] + h, x0[b] : x0[b] + w] of image b of the record image of [B, cin, Hin, Win] (y0, x0: HOST arrays
 *                                of B ints -- stacked tiles of one shape keep their own origins; beyond 8 images the origins must repeat every 8); outputs are [B, cout, 2h, 2w] (fp32 d_y
 *                                and / or record image d_y_rec with d_y_coef as in mdtile_conv2d_rec).  Window edges inside the image read the
 *                                image's real neighbours, so the outputs equal the same pixels of the whole-image call bit for bit. */
int mdtile_upconv2d_rec_window(const void* d_x_rec, const float* d_w_packed, const float* d_bias, float* d_y, void* d_y_rec,
                               const float* d_y_coef, int B, int cin, int cout, int Hin, int Win, const int* y0, const int* x0, int h, int w,
                               int flags, mdtile_stream_t stream);

/* Row-band pieces of get_var_mean (tilevae.py:207-215) for an activation that is split by rows across GPUs (sequence-parallel
 * fast-mode estimator): every plane holds plane_stride floats of which [offset, offset+len) are this rank's own rows.
 *   mdtile_gn_sums      : d_sums[B*groups][2] = fp64 (sum, sum of squares) over the own rows of each (sample, group);
 *                         d_ws: mdtile_gn_stats_ws_size(B, groups) bytes.  Ranks all-reduce(sum) d_sums.
 *   mdtile_gn_from_sums : mean = s1/count, var = s2/count - mean^2 (biased), count = elements per (sample, group) over ALL ranks */
int mdtile_gn_sums(const float* d_x, int B, int C, size_t plane_stride, size_t offset, size_t len, int groups, double* d_sums,
                   void* d_ws, mdtile_stream_t stream);
int mdtile_gn_from_sums(const double* d_sums, double count, int BG, float* d_mean, float* d_var, mdtile_stream_t stream);

/* attn_forward body between the 1x1 convs (tile_utils/attn.py:55-70): single head,
 * out[b,c,i] = sum_j v[b,c,j] * softmax_j(scale * sum_c' q[b,c',i] k[b,c',j]).
 * q,k: [B,C,T] channel-major;  v: [B,T,C] token-major (conv out_layout 1);  out: [B,C,T].  C = 128, 256 or 512.
 * Flash formulation (no T x T matrix).  Default: split-bf16 ("bf16x3") matrix-core kernel -- q, k, v and the
 * probabilities are split into bf16 hi/lo pairs (16 significand bits), fp32 accumulation and fp32 softmax statistics,
 * ~1e-5 relative to fp32; it stages fragment-order copies of q, k, v in d_ws (mdtile_vae_attn_ws_size(B,C,T) bytes).
 * MDTILE_ATTN_EXACT_F32 (or env MDTILE_ATTN_MODE=f32) selects the exact-fp32 MFMA kernel, which needs no workspace. */
#define MDTILE_ATTN_EXACT_F32 1
#define MDTILE_ATTN_V_CHANNEL_MAJOR 2   /* v is [B,C,T] like q and k (split-bf16 kernel only): the v projection then runs on the same 1x1 kernels */
size_t mdtile_vae_attn_ws_size(int B, int C, int T);
/* 1 when mdtile_vae_attn(C, flags) will run the split-bf16 kernel and therefore accepts MDTILE_ATTN_V_CHANNEL_MAJOR -- the SAME conditions
 * the dispatch inside mdtile_vae_attn tests (flags, mdtile_set_precision, env MDTILE_ATTN_MODE=f32, C); 0: hand v over token-major. */
int mdtile_vae_attn_takes_channel_major(int C, int flags);
int mdtile_vae_attn(const float* d_q, const float* d_k, const float* d_v, float* d_out, int B, int C, int T, float scale,
                    int flags, void* d_ws, mdtile_stream_t stream);
/* Same contraction with separate query / key token counts: q [B,C,Tq], k [B,C,Tk], v [B,Tk,C] -> out [B,C,Tq]
 * (a row band of queries against the keys / values gathered from every band).  Split-bf16 kernel; workspace as above. */
size_t mdtile_vae_attn_qk_ws_size(int B, int C, int Tq, int Tk);
int mdtile_vae_attn_qk(const float* d_q, const float* d_k, const float* d_v, float* d_out, int B, int C, int Tq, int Tk, float scale,
                       void* d_ws, mdtile_stream_t stream);

/* crop_valid_region + result[...] = tile (tilevae.py:248-259, 630-632): copies the valid window of one finished tile
 * [N,C,th,tw] into result [N,C,RH,RW].  in_bbox/out_bbox as returned by mdtile_vae_split_tiles. */
int mdtile_crop_store(const float* d_tile, int N, int C, int th, int tw, const int* in_bbox4, const int* out_bbox4,
                      int is_decoder, float* d_result, int RH, int RW, mdtile_stream_t stream);
/* tile extraction z[:, :, y1:y2, x1:x2] (tilevae.py:532-535) == mdtile_gather_rect with dtype f32 */

/* fast-mode estimator input (tilevae.py:545-559): scale_factor = tile_size / max(H, W); nearest-exact resample of
 * z [N,C,H,W] to [N,C,oh,ow] (sizes from mdtile_vae_fast_size), per-channel re-standardisation (unbiased std over
 * N,H,W of both), clamp to [min z, max z].  d_ws: mdtile_vae_fast_ws_size(C) bytes. */
int mdtile_vae_fast_size(int H, int W, int tile_size, int* oh, int* ow);
size_t mdtile_vae_fast_ws_size(int C);
int mdtile_vae_fast_input(const float* d_z, int N, int C, int H, int W, int tile_size, float* d_out, void* d_ws,
                          mdtile_stream_t stream);

/* ----------------------------------------------------------------------------------------------------------
 * Multi-GPU (upstream is single-device: scripts/tilediffusion.py:257-383 drives one `p`).
 * A shard context owns one RCCL communicator and one HIP stream per LOCAL rank.  RCCL is dlopen'ed on first use.
 *   mdtile_shard_init(ndev, dev_ids)      single process, one rank per listed device (ncclCommInitAll): usable from inside a webui.
 *                                         Listing a device twice (or MDTILE_SHARD_TRANSPORT=copy) selects the "copy" transport:
 *                                         hipMemcpyAsync + events instead of RCCL, same packing and summation (1-GPU functional runs)
 *   mdtile_shard_unique_id / _init_rank   one rank of a process-per-GPU job: rank 0 draws the 128-byte id, the host distributes it
 *   mdtile_shard_info                     info4 = { ranks, local ranks, first local rank, 1 = RCCL | 0 = copy transport }
 * Per-rank arrays below have one entry per LOCAL rank; streams == NULL uses the context's own streams (mdtile_shard_stream).
 *   mdtile_halo_exchange   band_rows[2 r], [2 r + 1] = canvas rows [lo, hi) the tiles of rank r touch (contiguous bands of tile rows).
 *                          d_partial[i] [N,C,H,W] fp32 holds rank i's partial sums (mdtile_blend with MDTILE_BLEND_PARTIAL); rows shared
 *                          with other bands are packed, swapped (grouped ncclSend / ncclRecv) and summed in ascending rank order on
 *                          every side (bit-identical everywhere); then mdtile_blend_finalize.  d_scratch[i]: mdtile_halo_scratch_bytes.
 *   mdtile_allreduce_stats all-reduce(sum) of `count` doubles in place (slow-mode GroupNorm pooling across ranks, tilevae.py:320-335)
 *   mdtile_shard_bcast     `bytes` from rank `root` to all ranks (a region's model output to the bands that composite it)
 *   mdtile_shard_p2p       grouped point-to-point (one ncclGroup per call): ops[i][0 .. n_ops[i]) are local rank i's transfers; an op
 *                          sends and / or receives (bytes == 0 skips that half); the k-th send of a to b pairs with b's k-th receive
 *                          from a.  Carries the row halos of the sequence-parallel estimator (both directions in one group) and the
 *                          gather of the decoded tile rectangles to the rank that returns the image
 *   mdtile_shard_allgather every rank contributes `bytes`; d_recv[i] (nranks * bytes) holds them in rank order (K / V of the
 *                          sequence-parallel estimator's attention, tile_utils/attn.py:55-67)
 *   mdtile_shard_selfcheck bring-up check: all-reduce, broadcast, grouped ring send / receive (a self send / receive on one rank) and
 *                          all-gather of small payloads, verified on the host; d_scratch[i] >= mdtile_shard_selfcheck_bytes(sh).
 *                          MDTILE_SHARD_TRANSPORT=rccl makes a one-device context use a 1-rank RCCL communicator (it needs none).
 *   mdtile_shard_probe_rank  interruptible bring-up probe of a process-per-GPU communicator: the same rendezvous as mdtile_shard_init_rank on a
 *                          NON-BLOCKING communicator, polled until it is up or `timeout_s` has passed, then aborted (ncclCommAbort) -- the calling
 *                          thread is never left blocked inside RCCL.  Every rank calls it with the same id (an id of its own, not the one the
 *                          real communicator will use).  MDTILE_OK: the rendezvous works, go on to mdtile_shard_init_rank. */
typedef struct mdtile_shard mdtile_shard;
mdtile_shard* mdtile_shard_init(int ndev, const int* dev_ids);
int mdtile_shard_unique_id(void* id128);
mdtile_shard* mdtile_shard_init_rank(int nranks, int rank, const void* id128, int device);
int mdtile_shard_probe_rank(int nranks, int rank, const void* id128, int device, double timeout_s);
void mdtile_shard_destroy(mdtile_shard* sh);
int mdtile_shard_info(const mdtile_shard* sh, int* info4);
mdtile_stream_t mdtile_shard_stream(const mdtile_shard* sh, int local_rank);
size_t mdtile_halo_scratch_bytes(int nranks, int rank, const int* band_rows, int N, int C, int W);
int mdtile_halo_exchange(mdtile_shard* sh, float* const* d_partial, void* const* d_scratch, int N, int C, int H, int W,
                         const int* band_rows, const mdtile_stream_t* streams);
int mdtile_allreduce_stats(mdtile_shard* sh, double* const* d_buf, int count, const mdtile_stream_t* streams);
int mdtile_shard_bcast(mdtile_shard* sh, void* const* d_buf, size_t bytes, int root, const mdtile_stream_t* streams);
typedef struct mdtile_p2p {
    int peer;
    const void* send;
    size_t send_bytes;
    void* recv;
    size_t recv_bytes;
} mdtile_p2p;
int mdtile_shard_p2p(mdtile_shard* sh, const mdtile_p2p* const* ops, const int* n_ops, const mdtile_stream_t* streams);
int mdtile_shard_allgather(mdtile_shard* sh, const void* const* d_send, void* const* d_recv, size_t bytes, const mdtile_stream_t* streams);
size_t mdtile_shard_selfcheck_bytes(const mdtile_shard* sh);
int mdtile_shard_selfcheck(mdtile_shard* sh, void* const* d_scratch, const mdtile_stream_t* streams);

#ifdef __cplusplus
}
#endif
#endif /* MDTILE_H */
