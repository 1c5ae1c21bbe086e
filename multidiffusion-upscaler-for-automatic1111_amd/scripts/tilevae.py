"""
Tiled VAE on the mdtile engine -- same plugin surface as upstream scripts/tilevae.py:
`Script.title/show/ui/process/postprocess` with the positional argument order
(enabled, encoder_tile_size, decoder_tile_size, vae_to_gpu, fast_decoder, fast_encoder, color_fix), and a `VAEHook`
callable that replaces `vae.decoder.forward` (upstream :739-745).

What changed underneath (MI355X-first, 288 GB HBM):
  * the ldm Decoder is compiled ONCE into a flat program (upstream's task queue, :107-195) whose steps are mdtile C-ABI
    calls: fp32-MFMA implicit-GEMM convs with fused residual add / nearest-2x upsample, fixed-statistics
    GroupNorm+SiLU in one pass, a flash-style attention kernel (no T x T matrix), crop+store into the result;
  * tiles, residuals and parked activations never leave the GPU (upstream's .cpu()/.to(device) ping-pong, :534-642,
    exists only to fit small VRAM);
  * GroupNorm semantics are upstream's: statistics frozen from a down-sampled latent in fast mode (:542-563, :464-505)
    or pooled across tiles at every norm in slow mode (:320-335) -- NOT the untiled network's.
Both directions run on the engine: the decoder (latent -> image, pad 11) and the encoder (image -> latent moments, pad 32,
stride-2 `Downsample` convs, optional `color_fix` semi-fast mode: statistics frozen only up to the first downsample,
upstream :492-496).
"""
from __future__ import annotations

from time import time
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

import modules.scripts as scripts
import modules.devices as devices
from modules.shared import state

import mdtile


import os as _os
# GroupNorm + SiLU fused into the following 3x3 conv's input staging (engine: mdtile_conv2d_gn).  MDTILE_FUSE_GN=0 keeps the
# separate one-pass GroupNorm+SiLU kernel (A/B measurements, debugging).
FUSE_PRE_GN = _os.environ.get("MDTILE_FUSE_GN", "1") != "0"
# fast mode with every norm frozen: activations travel between the 3x3 convs as split-bf16 record images that the PRODUCING
# conv writes already normalised + SiLU'd (engine: mdtile_conv2d_rec); MDTILE_REC=0 keeps the fp32 hand-over (A/B, debugging)
TILE_BATCH = int(_os.environ.get("MDTILE_TILE_BATCH", "4"))     # fast mode: tiles of equal shape per sweep (see vae_tile_forward); 4 = the four
# corners / top-bottom edges / left-right edges / interior tiles of a 4 x 4 grid each go as ONE stack (3 left a single-tile sweep per group: profiles/r5d)
REC_PATH = _os.environ.get("MDTILE_REC", "1") != "0"
# slow mode and the estimator pass: the record kernels at the pooled-statistics sites where they pay (VAEHook._pooled_site_takes_rec); 0 = fp32 hand-over
SLOW_REC = _os.environ.get("MDTILE_SLOW_REC", "1") != "0"
# slow mode: the conv that produces a pooled norm's input leaves that input's (var, mean) from its own epilogue where such a kernel exists
# (PackedConv.leaves_stats); 0 = a statistics pass over every tile at every norm (A/B, debugging)
SLOW_STATS = _os.environ.get("MDTILE_SLOW_STATS", "1") != "0"
# fast-mode decoder tiles shed their dead border where the resolution doubles (live_windows below); 0 = decode the whole padded tile
LIVE_WINDOW = _os.environ.get("MDTILE_LIVE_WINDOW", "1") != "0"
# multi-GPU fast mode: run the GroupNorm estimator sequence-parallel across the ranks (mdtile/seqpar.py); 0 = every rank
# repeats the whole estimator (no communication, but 1 of every rank's ~3 work units at 8 GPUs)
SP_ESTIMATOR = _os.environ.get("MDTILE_SP_ESTIMATOR", "1") != "0"


def get_rcmd_enc_tsize() -> int:
    """Upstream picks by VRAM (:79-87); every MI355X has 288 GB, i.e. the top bucket."""
    return 3072 if torch.cuda.is_available() else 512


def get_rcmd_dec_tsize() -> int:
    """Upstream: 256 for > 30 GB, 64 off-GPU (:90-99)."""
    return 256 if torch.cuda.is_available() else 64


# the program (task queue with the engine's fusions), live-window arithmetic, crop_valid_region, GroupNormParam, TileState
from tile_utils.vae_program import (AttnPack, GroupNormParam, Step, TileState, _norm_params, _pack, _resblock,      # noqa: E402,F401
                                    build_task_queue, crop_valid_region, live_windows)


# ---------------------------------------------------------------------------------------------------------------------
# GroupNorm statistics (upstream :207-245; the slow-mode collector GroupNormParam, :289-361, lives in tile_utils/vae_program.py)
# ---------------------------------------------------------------------------------------------------------------------
def get_var_mean(input: Tensor, num_groups: int, eps: float = 1e-6) -> Tuple[Tensor, Tensor]:
    return mdtile.gn_stats(input, num_groups)


def custom_group_norm(input, num_groups, mean, var, weight=None, bias=None, eps=1e-6, silu: bool = False):
    return mdtile.gn_apply(input, mean, var, weight, bias, num_groups, eps, silu)


# ---------------------------------------------------------------------------------------------------------------------
class VAEHook:

    def __init__(self, net, tile_size, is_decoder: bool, fast_decoder: bool, fast_encoder: bool, color_fix: bool,
                 to_gpu: bool = False):
        # (signature == upstream's, :364-372.)  engine / _pack / _sp_ops: the mdtile module, the conv packer and the
        # sequence-parallel ops the hook talks to.  The product leaves the defaults (the HIP engine, GPU only); the CPU tests of
        # this host logic overwrite the three attributes with torch doubles (tests/torch_engine.py).
        self.engine = mdtile
        self._pack = None
        self._sp_ops = None
        self.net = net
        self.tile_size = tile_size
        self.is_decoder = is_decoder
        self.fast_mode = (fast_encoder and not is_decoder) or (fast_decoder and is_decoder)
        self.color_fix = color_fix and not is_decoder
        self.to_gpu = to_gpu
        self.pad = 11 if is_decoder else 32
        self._program: Optional[List[Step]] = None
        self.last_seconds = None
        self.shard = (0, 1)   # (rank, world): process-per-GPU runs decode tiles rank, rank+world, ... (mdtile/sharding.py)
        # process-per-GPU runs: the rank whose call returns the ASSEMBLED image, as upstream's single tensor (:630-656) -- the other
        # ranks' tile rectangles travel to it in one grouped exchange; None leaves every rank with only its own tiles filled in
        self.gather_to: Optional[int] = None
        # single-process multi-device decode (what a webui process can use): CUDA device indices, e.g. [0, 1, 2, 3]; the tiles are
        # dealt to the devices by area, each with its own copy of the packed weights; fast mode only (no collective needed:
        # the frozen statistics are computed once and copied).  A device may be listed twice (functional runs on one GPU).
        self.devices: Optional[List[int]] = None
        self._dev_programs = {}

    def __call__(self, x):
        original_device = next(self.net.parameters()).device
        try:
            if self.to_gpu:
                self.net = self.net.to(devices.get_optimal_device())
            B, C, H, W = x.shape
            if max(H, W) <= self.pad * 2 + self.tile_size:
                print("[Tiled VAE]: the input size is tiny and unnecessary to tile.")
                return self.net.original_forward(x)
            return self.vae_tile_forward(x)
        finally:
            self.net = self.net.to(original_device)

    # ---- geometry (host ints via the C ABI) -------------------------------------------------------------------------
    def get_best_tile_size(self, lowerbound, upperbound):
        """Upstream's helper (:390-403): the smallest multiple of 32 / 16 / 8 / 4 / 2 above `lowerbound` that still fits under
        `upperbound`.  The split itself lives behind the C ABI (mdtile_vae_split_tiles uses the same function)."""
        return self.engine.vae_best_tile_size(lowerbound, upperbound)

    def split_tiles(self, h, w):
        return self.engine.vae_split_tiles(h, w, self.tile_size, self.is_decoder)

    # ---- program ------------------------------------------------------------------------------------------------------
    def program(self) -> List[Step]:
        dev = next(self.net.parameters()).device
        if self._program is None or self._program_dev != dev:
            self._program = build_task_queue(self.net, self.is_decoder, self._pack, self.engine)
            self._program_dev = dev
        return self._program

    @staticmethod
    def _feeds_norm(steps: List[Step], i: int) -> bool:
        """The value steps[i] produces is the input of a GroupNorm (only residual bookkeeping in between)."""
        j = i + 1
        while j < len(steps) and steps[j].kind == "store_res":
            j += 1
        return j < len(steps) and steps[j].kind == "norm"

    def _run_until_norm(self, steps: List[Step], st: TileState, want_stats: bool = False):
        """Advance one tile to its next GroupNorm (exclusive) or to the end.
        want_stats (slow mode, the norm ahead is pooled): the conv that produces the norm's input also leaves its (var, mean) in st.stats
        where a kernel does that in its epilogue (PackedConv.leaves_stats) -- GroupNormParam.add_tile then needs no pass over the tile."""
        while st.pc < len(steps):
            s = steps[st.pc]
            if s.kind == "norm":
                return
            if s.kind != "store_res":
                st.stats = None
            if s.kind == "store_res":
                st.res.append(st.x if s.conv is None else s.conv(st.x))
            elif s.kind == "conv":
                route = self._conv_route(s, st.pre is not None, want_stats and self._feeds_norm(steps, st.pc))
                residual = st.res.pop() if s.fuse_res and route != "down" else None
                if route == "down":
                    st.x = s.conv.down2(st.x)
                elif route == "rec_stats":
                    xrec = self.engine.rec_from_f32(st.x, st.pre)
                    st.x, st.stats = s.conv.call_rec_stats(xrec, residual=residual, upsample2x=s.upsample, groups=32)
                elif route == "handover_stats":
                    st.x, st.stats = s.conv.call_stats(st.x, st.pre, residual=residual, groups=32)
                elif route == "rec":
                    # pooled-statistics site on the record kernels: one conversion pass (norm + SiLU fused into it) + the record conv --
                    # the fp32 hand-over kernel's output to fp32 rounding (a fused residual enters the accumulation first here: (res + sum) + bias), cheaper where _pooled_site_takes_rec says so
                    xrec = self.engine.rec_from_f32(st.x, st.pre)
                    st.x, _ = s.conv.call_rec(xrec, residual=residual, upsample2x=s.upsample, want_f32=True, want_rec=False)
                else:
                    st.x = s.conv(st.x, residual=residual, upsample2x=s.upsample, pre_gn=st.pre)
                st.pre = None
            elif s.kind == "attn":
                st.x = s.attn(st.x, st.res.pop())
            elif s.kind == "tanh":
                st.x = self.engine.tanh(st.x)
            st.pc += 1

    def _conv_route(self, s: Step, has_pre: bool, want_stats: bool = False) -> str:
        """Which engine call a conv step of the norm-to-norm walk (slow mode, the estimator pass) takes -- THE one place that decides it:
        _run_until_norm dispatches on it and _apply_norm asks it whether the norm it has just resolved may ride on the conv as pending
        (a, s) coefficients (`has_pre`), so the two cannot disagree about who applies a norm.
          "down"            ldm Downsample (stride 2; no norm in front of it)
          "rec_stats"       conversion pass (takes the pending coefficients) + record conv that leaves get_var_mean of its output
          "handover_stats"  fp32 hand-over conv applying the pending coefficients while staging + statistics from its epilogue
          "rec"             conversion pass + record conv (_pooled_site_takes_rec: where that is the cheaper pair)
          "plain"           the conv's own call: applies pending coefficients only where fuses_pre_gn() says it can"""
        if s.downsample:
            return "down"
        takes_rec = self._pooled_site_takes_rec(s)
        stats_fn = getattr(s.conv, "leaves_stats", None) if (want_stats and SLOW_STATS) else None
        if stats_fn is not None and takes_rec and stats_fn(32, upsample2x=s.upsample, rec=True):
            return "rec_stats"
        if stats_fn is not None and has_pre and not s.upsample and not takes_rec and stats_fn(32):
            return "handover_stats"
        return "rec" if takes_rec else "plain"

    def _norm_rides_on(self, nxt: Optional[Step]) -> bool:
        """A resolved norm + SiLU in front of `nxt` stays pending as (a, s) coefficients exactly when the route `nxt` will take applies them."""
        if not (FUSE_PRE_GN and nxt is not None and nxt.kind == "conv" and not nxt.downsample):
            return False
        route = self._conv_route(nxt, True)        # (statistics variants of a route apply the coefficients the same way)
        return (route == "rec" and not nxt.upsample) or (route == "plain" and bool(nxt.conv.fuses_pre_gn(upsample2x=nxt.upsample)))

    def _pooled_site_takes_rec(self, s: Step) -> bool:
        """Slow mode / the estimator pass: a norm whose statistics are pooled cannot be applied by the conv that PRODUCES its input, so the
        record kernels cost an extra conversion pass (fp32 -> activated records) there.  They still win where the conv is long against its
        activation: the 512 -> 512 layers (-7 ... -9 % incl. the pass) and every upsample conv (the pass runs on the quarter-size input:
        -8 ... -21 %); at 256 / 128 input channels the pass costs more than the faster conv saves (+1 ... +8 %) -- profiles/r4z/conv_probe.log.
        conv_out (cout < 32) behind a pooled norm_out: no hand-over kernel applies a norm for so few couts, so the alternative is a norm pass
        (1R + 1W) + the exact-fp32 conv -- 3.8 ms per 2224^2 tile against 1.9 ms for conversion pass + narrow record conv (profiles/r5q:
        kernel_stats_slow.csv, 61 ms of a slow-mode 8K decode)."""
        if not (SLOW_REC and REC_PATH and hasattr(self.engine, "rec_from_f32") and self._takes_rec(s)):
            return False
        c = s.conv
        return bool(s.upsample or (getattr(c, "cin", 0) >= 512 and getattr(c, "cout", 0) >= 512) or 0 < getattr(c, "cout", 0) < 32)

    def _tile_batch_that_fits(self, N: int, tile_hw: Tuple[int, int], dev) -> int:
        """Tiles of one shape per sweep (TILE_BATCH at most): what 60 % of the free device memory holds.  Peak of one tile: about five
        live tensors (residual, fp32 activation, two record images, conv output) of 4 B x 128 channels at the widest level -- 8x the
        latent tile for the decoder, the image tile itself for the encoder."""
        if TILE_BATCH <= 1 or torch.device(dev).type != "cuda":
            return max(1, TILE_BATCH)
        h, w = tile_hw
        px = h * w * (64 if self.is_decoder else 1)
        per_tile = 5 * 4 * 128 * px * N
        free, _total = torch.cuda.mem_get_info(dev)
        # blocks torch's caching allocator holds but has handed to nobody are as good as free for the next sweep (after the first decode,
        # or a UNet run, the driver-level figure alone can be tiny on cards smaller than the 288 GB this was developed on)
        free += max(0, torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))
        tb = max(1, min(TILE_BATCH, int(0.6 * free // max(per_tile, 1))))
        if tb < TILE_BATCH and not getattr(self, "_said_tb", False):
            self._said_tb = True
            print(f"[Tiled VAE]: {tb} instead of {TILE_BATCH} tiles per sweep ({free / 2**30:.1f} GiB free, {per_tile / 2**30:.1f} GiB per {h}x{w} tile)")
        return tb

    # ---- fast mode, every norm frozen: record-image hand-over between the 3x3 convs ----------------------------------
    @staticmethod
    def _takes_rec(step: Step) -> bool:
        fn = getattr(step.conv, "takes_rec", None)
        return bool(fn and not step.downsample and fn(step.upsample))

    def _demand(self, steps: List[Step], i: int):
        """Forms in which the value produced by steps[i] has to exist: (fp32 NCHW?, record image: None | "raw" | index of the
        norm step whose (a, s) + SiLU the producer applies)."""
        need_f32, j = False, i + 1
        while j < len(steps) and steps[j].kind == "store_res":     # residual / nin_shortcut input: fp32
            need_f32, j = True, j + 1
        if j >= len(steps):
            return True, None
        s = steps[j]
        if s.kind == "norm":
            nxt = steps[j + 1] if j + 1 < len(steps) else None
            if s.silu and nxt is not None and nxt.kind == "conv" and not nxt.upsample and self._takes_rec(nxt):
                return need_f32, j
            return True, None
        if s.kind == "conv" and s.upsample and self._takes_rec(s):
            return need_f32, "raw"
        return True, None

    def _run_tile_rec(self, steps: List[Step], x: Tensor, frozen, coefs, norm_ord, windows=None, first: int = 0, last: Optional[int] = None,
                      xrec=None):
        """One tile start to finish with frozen statistics (upstream's single sweep, :578-642).  A 3x3 conv that the record
        kernels take reads its input as a record image; whoever produces that input writes it in that form -- the previous
        record conv's epilogue (norm + SiLU + split fused), or mdtile_rec_from_f32 behind conv_in / attention.
        windows (live_windows): the upsample convs listed there compute only that window of their input plane.
        first / last / xrec: run steps[first:last] only; a sweep cut in front of an upsample conv (between two resblocks: no residual is
        pending there) hands over (x, xrec) -- returned instead of x when `last` is given -- and is resumed with them."""
        E = self.engine
        res: List[Tensor] = []
        pre = None
        stop = len(steps) if last is None else last
        for i in range(first, stop):
            s = steps[i]
            if s.kind == "store_res":
                res.append(x if s.conv is None else s.conv(x))
            elif s.kind == "norm":
                k = norm_ord[i]
                nxt = steps[i + 1] if i + 1 < len(steps) else None
                is_conv = nxt is not None and nxt.kind == "conv" and not nxt.downsample
                if s.silu and is_conv and not nxt.upsample and self._takes_rec(nxt):
                    if xrec is None:
                        xrec = E.rec_from_f32(x, coefs[k])
                elif FUSE_PRE_GN and s.silu and is_conv and nxt.conv.fuses_pre_gn(upsample2x=nxt.upsample):
                    pre = coefs[k]
                else:
                    var, mean = frozen[k]
                    keep = res and res[-1] is x
                    x = E.gn_apply(x, mean, var, s.norm[0], s.norm[1], 32, 1e-6, s.silu, out=None if keep else x)
            elif s.kind == "conv":
                if s.downsample:
                    x, xrec = s.conv.down2(x), None
                else:
                    residual = res.pop() if s.fuse_res else None
                    if self._takes_rec(s) and (xrec is not None or s.upsample):
                        if xrec is None:
                            xrec = E.rec_from_f32(x, None)
                        need_f32, rk = self._demand(steps, i)
                        win = windows.get(i) if windows else None
                        x, xrec = s.conv.call_rec(xrec, residual=residual, upsample2x=s.upsample, want_f32=need_f32, want_rec=rk is not None,
                                                  rec_coef=None if rk in (None, "raw") else coefs[norm_ord[rk]], **({"window": win} if win else {}))
                    elif s.upsample and windows and windows.get(i):
                        assert residual is None and pre is None      # ldm Upsample: no norm in front, no skip connection into it
                        x, xrec = self._upconv_window_f32(s.conv, x, windows[i]), None
                    else:
                        x, xrec = s.conv(x, residual=residual, upsample2x=s.upsample, pre_gn=pre), None
                pre = None
            elif s.kind == "attn":
                x, xrec = s.attn(x, res.pop()), None
            elif s.kind == "tanh":
                x = E.tanh(x)
        if last is not None:
            assert not res and pre is None, "a sweep can only be cut between two resblocks"
            return x, xrec
        return x

    @staticmethod
    def _upconv_window_f32(conv, x: Tensor, window) -> Tensor:
        """The window form of an upsample conv that is NOT on the record kernels (exact-fp32 mode, channel counts they do not take): the
        fp32 hand-over kernel over the window plus the conv's own 1 px halo of real neighbours, the halo's outputs cut off afterwards --
        per kept pixel the same arithmetic as the whole-plane call (mdtile_upconv2d_rec_window does this without the two copies)."""
        B, _, H, W = x.shape
        y0s, x0s, h, w = window
        y0s = [int(y0s)] * B if isinstance(y0s, int) else [int(v) for v in y0s]
        x0s = [int(x0s)] * B if isinstance(x0s, int) else [int(v) for v in x0s]
        out, b0 = None, 0
        while b0 < B:                                     # runs of images with one origin (the N latents of one stacked tile)
            b1 = b0 + 1
            while b1 < B and (y0s[b1], x0s[b1]) == (y0s[b0], x0s[b0]):
                b1 += 1
            y0, x0 = y0s[b0], x0s[b0]
            ya, yb, xa, xb = max(y0 - 1, 0), min(y0 + h + 1, H), max(x0 - 1, 0), min(x0 + w + 1, W)
            y = conv(x[b0:b1, :, ya:yb, xa:xb].contiguous(), upsample2x=True)
            oy, ox = 2 * (y0 - ya), 2 * (x0 - xa)
            if out is None:
                out = torch.empty((B, y.shape[1], 2 * h, 2 * w), dtype=y.dtype, device=y.device)
            out[b0:b1] = y[:, :, oy:oy + 2 * h, ox:ox + 2 * w]
            b0 = b1
        return out

    def _live_plan(self, steps: List[Step], in_bbox, out_bbox):
        """(windows, narrowed input bbox) of one decoder tile for _run_tile_rec / crop_store -- ({}, in_bbox) when live-window narrowing does
        not apply (switched off, encoder)."""
        if not (LIVE_WINDOW and self.is_decoder):
            return {}, tuple(in_bbox)
        x1, x2, y1, y2 = in_bbox
        ox1, ox2, oy1, oy2 = out_bbox
        if any(v % 8 for v in (ox1, ox2, oy1, oy2)):
            return {}, tuple(in_bbox)
        valid = (oy1 // 8 - y1, ox1 // 8 - x1, oy2 // 8 - y1, ox2 // 8 - x1)
        windows, (ry0, rx0, ry1, rx1) = live_windows(steps, (y2 - y1, x2 - x1), valid)
        return windows, (x1 + rx0, x1 + rx1, y1 + ry0, y1 + ry1)

    def _apply_norm(self, steps: List[Step], st: TileState, var: Tensor, mean: Tensor):
        E = self.engine
        s = steps[st.pc]
        gamma, beta = s.norm
        nxt = steps[st.pc + 1] if st.pc + 1 < len(steps) else None
        if s.silu and self._norm_rides_on(nxt):
            # norm + SiLU ride on the conv's input staging: only the per-channel (a, s) pair is formed here
            st.pre = E.gn_coeffs(mean, var, gamma, beta, st.x.shape[1], 32, 1e-6)
        else:
            keep = st.res and st.res[-1] is st.x          # identity shortcut: the residual aliases the pre-norm tensor
            st.x = E.gn_apply(st.x, mean, var, gamma, beta, 32, 1e-6, s.silu, out=None if keep else st.x)
            st.stats = None                               # (var, mean) described the tensor that was just replaced
        st.pc += 1

    @torch.no_grad()
    def estimate_group_norm(self, z: Tensor, steps: List[Step]) -> Optional[List[Tuple[Tensor, Tensor]]]:
        """Fast mode: run the program on the down-sampled latent and freeze (var, mean) at every norm (:464-505).
        Returns None (-> slow mode) if a NaN shows up, as upstream."""
        st = TileState(z)
        frozen = []
        n_norm = sum(1 for s in steps if s.kind == "norm")
        if self.color_fix:
            # semi-fast encoder mode (upstream :492-496): the estimate stops at the first downsample; only the norms before
            # it are frozen, the others are pooled across the tiles as in slow mode
            first_down = next((i for i, s in enumerate(steps) if s.kind == "conv" and s.downsample), len(steps))
            n_norm = sum(1 for s in steps[:first_down] if s.kind == "norm")
        while True:
            self._run_until_norm(steps, st, want_stats=True)
            if st.pc >= len(steps):
                break
            # (the conv that produced st.x has left its statistics where a kernel does that in its epilogue: TileState.stats)
            var, mean = st.stats if st.stats is not None else self.engine.gn_stats(st.x, 32)
            frozen.append((var, mean))
            if len(frozen) == n_norm:
                break
            self._apply_norm(steps, st, var, mean)
        # upstream tests the activation for NaN after every norm (:487-490) and falls back to slow mode; a NaN anywhere upstream of a
        # norm poisons that norm's statistics, so ONE test of the frozen (var, mean) rows at the end sees the same events without a
        # host sync per norm
        if frozen and bool(torch.isnan(torch.stack([v.sum() + m.sum() for v, m in frozen])).any().item()):
            print("Nan detected in fast mode estimation. Fast mode disabled.")
            return None
        return frozen

    def _pooled_across_ranks(self, gp: "GroupNormParam", steps, dev, interrupted: bool = False):
        """Slow mode on several GPUs: all-reduce(sum) of [sum px*mean, sum px*var, sum px] (2*B*32+1 floats) per barrier, on the
        job's data plane (the engine's RCCL communicator when the process has one, mdtile/sharding.py).
        Returns (pooled or None, any rank interrupted).  The interrupt rides in the `head` exchange every rank enters at every pooled
        barrier: a rank that saw state.interrupted keeps walking the barriers (with no tiles) until it has said so HERE, and all ranks
        leave the lockstep loop together -- a rank that simply left would pair its next collective (the 2-float agreement in front of
        the gather) with its peers' `head` and hang them in allreduce_stats."""
        from mdtile import sharding
        BG = None
        if gp.var_list and not interrupted:
            px = torch.tensor(gp.pixel_list, dtype=torch.float32, device=dev).unsqueeze(1)
            sm, sv, sp = (torch.vstack(gp.mean_list) * px).sum(0), (torch.vstack(gp.var_list) * px).sum(0), px.sum().view(1)
            BG = sm.numel()
        # who still has tiles at this barrier, and how wide the statistics rows are (ranks without tiles contribute zeros)
        head = sharding.comm_allreduce_sum(torch.tensor([1.0 if BG else 0.0, float(BG or 0), 1.0 if interrupted else 0.0], dtype=torch.float64, device=dev))
        if head[2].item() > 0.0:
            return None, True
        if head[0].item() == 0.0:
            return None, False
        if BG is None:
            BG = int(round(head[1].item() / head[0].item()))
            sm, sv, sp = torch.zeros(BG, device=dev), torch.zeros(BG, device=dev), torch.zeros(1, device=dev)
        return sharding.allreduce_stats(sm, sv, sp), False

    # ---- single-process multi-device sweep ---------------------------------------------------------------------------------
    def _program_on(self, index: int) -> List[Step]:
        """The task queue with its weights packed on CUDA device `index` (built once per device)."""
        import copy
        if index not in self._dev_programs:
            d = torch.device("cuda", index)
            src = next(self.net.parameters()).device
            net = self.net if src == d else copy.deepcopy(self.net).to(d)
            with torch.cuda.device(d):
                self._dev_programs[index] = build_task_queue(net, self.is_decoder, self._pack, self.engine)
        return self._dev_programs[index]

    def _multi_device_sweep(self, z: Tensor, steps0: List[Step], frozen, in_bboxes, out_bboxes, dtype, t0) -> Tensor:
        """Fast mode on several devices of ONE process: the tiles are dealt to the devices by area (mdtile/sharding.py: deal_tiles, the deal of
        the process-per-GPU path), each device with its own stream, its own packed weights and a copy of the frozen statistics; launches
        are asynchronous, so one Python thread keeps all devices busy -- the tiles are issued device by device in rounds (every device's
        first tile, then every device's second ...), so no device waits for the host to finish another device's list.  The output tiles
        are disjoint (out_bboxes never overlap): each device crops into its own canvas and its rectangles are copied to the first
        device at the end (peer copies over xGMI)."""
        E = self.engine
        N, _, height, width = z.shape
        devs = [torch.device("cuda", i) for i in self.devices]
        norm_idx = [i for i, s in enumerate(steps0) if s.kind == "norm"]
        norm_ord = {i: k for k, i in enumerate(norm_idx)}
        per = []
        for d in devs:
            with torch.cuda.device(d):
                steps = self._program_on(d.index)
                fz = [(v.to(d), m.to(d)) for (v, m) in frozen]
                coefs = [E.gn_coeffs(m, v, steps[i].norm[0], steps[i].norm[1], steps[i].channels, 32, 1e-6) for i, (v, m) in zip(norm_idx, fz)]
                per.append(dict(steps=steps, frozen=fz, coefs=coefs, z=z.to(d), result=None, flags=[], mine=[]))
        from mdtile import sharding as _sh
        owner = _sh.deal_tiles(in_bboxes, len(devs))
        lists = [[i for i in range(len(in_bboxes)) if owner[i] == k] for k in range(len(devs))]
        issue = [(lst[r], k) for r in range(max(len(lst) for lst in lists)) for k, lst in enumerate(lists) if r < len(lst)]
        for i, k in issue:
            if state.interrupted:
                break
            L, d = per[k], devs[k]
            with torch.cuda.device(d):
                b = in_bboxes[i]
                x = E.gather_rect(L["z"], b[0], b[2], b[1] - b[0], b[3] - b[2])
                windows, live_bbox = self._live_plan(L["steps"], in_bboxes[i], out_bboxes[i])
                x = self._run_tile_rec(L["steps"], x, L["frozen"], L["coefs"], norm_ord, windows)
                if L["result"] is None:
                    oh, ow = (height * 8, width * 8) if self.is_decoder else (height // 8, width // 8)
                    L["result"] = torch.zeros((N, x.shape[1], oh, ow), device=d, dtype=torch.float32)
                L["flags"].append(torch.isnan(x).all())
                E.crop_store(x, live_bbox, out_bboxes[i], L["result"], self.is_decoder)
                L["mine"].append(i)
        if all(L["result"] is None for L in per):
            # interrupted before any tile finished: as the single-device path (and upstream, :644-650)
            if not self.is_decoder:
                raise RuntimeError("[Tiled VAE]: interrupted before any encoder tile finished")
            from modules.sd_vae_approx import cheap_approximation
            return torch.cat([torch.nn.functional.interpolate(cheap_approximation(x).unsqueeze(0), scale_factor=8, mode="nearest-exact")
                              for x in z], dim=0).to(devs[0], dtype=dtype)
        if per[0]["result"] is None:      # the first device finished nothing (interrupt): it still hosts the assembled canvas
            with torch.cuda.device(devs[0]):
                ref_res = next(L["result"] for L in per if L["result"] is not None)
                per[0]["result"] = torch.zeros(ref_res.shape, device=devs[0], dtype=torch.float32)
        out = per[0]["result"]             # interrupted runs return the finished tiles, like upstream
        bad = False
        for k, L in enumerate(per):
            torch.cuda.current_stream(devs[k]).synchronize()
            bad = bad or (L["flags"] and bool(torch.stack(L["flags"]).any().item()))
            if k == 0 or L["result"] is None:
                continue
            for i in L["mine"]:
                x1, x2, y1, y2 = out_bboxes[i]
                out[:, :, y1:y2, x1:x2].copy_(L["result"][:, :, y1:y2, x1:x2])
        if bad:
            devices.test_for_nans(torch.full((1,), float("nan")), "vae")
        self.last_seconds = time() - t0
        torch.cuda.synchronize(devs[0])
        print(f"[Tiled VAE]: Done in {time() - t0:.3f}s on {len(devs)} devices")
        return out.to(dtype)

    # ---- the tile sweep (upstream vae_tile_forward, :507-656) ------------------------------------------------------------------------
    # vae_tile_forward is the driver: split, estimator, deal the tiles, run ONE of three sweeps, assemble.
    #   _sweep_frozen      every norm frozen (fast mode): each tile runs start to finish on its own -- stacked by shape on the record path
    #   _sweep_lockstep    slow mode / color_fix: all tiles advance from norm to norm, statistics pooled at each one
    #   _multi_device_sweep  fast mode on several devices of one process (above)
    class _Run:
        """State of one vae_tile_forward call shared by the sweeps: this rank's tiles, the result canvas, the NaN flags, the live windows."""

        def __init__(self, hook, z, in_bboxes, out_bboxes, mine):
            self.hook, self.z, self.in_bboxes, self.out_bboxes, self.mine = hook, z, in_bboxes, out_bboxes, list(mine)
            E = hook.engine
            sel = set(mine)
            self.tiles = {i: TileState(E.gather_rect(z, b[0], b[2], b[1] - b[0], b[3] - b[2])) for i, b in enumerate(in_bboxes) if i in sel}
            self.result = None
            self.nan_flags = []
            self.live = {}          # tile -> (windows of its upsample convs, the input bbox of what is left of it): live_windows
            self.interrupted = False

        def finish(self, i: int):
            """crop_valid_region + `result[...] = tile` of one finished tile (upstream :630-632); the canvas appears with the first one."""
            hook, E, z = self.hook, self.hook.engine, self.z
            x = self.tiles[i].x
            if self.result is None:
                N, _, height, width = z.shape
                oh, ow = (height * 8, width * 8) if hook.is_decoder else (height // 8, width // 8)
                self.result = torch.zeros((N, x.shape[1], oh, ow), device=z.device, dtype=torch.float32)
            self.nan_flags.append(torch.isnan(x).all())       # upstream tests every tile (:626); here ONE host read per decode
            E.crop_store(x, self.live[i][1] if i in self.live else self.in_bboxes[i], self.out_bboxes[i], self.result, hook.is_decoder)
            self.tiles[i] = None

    def _sweep_frozen(self, run: "VAEHook._Run", steps: List[Step], frozen) -> None:
        """Every norm is already resolved: each tile runs start to finish on its own (upstream: one sweep, :578-642)."""
        E, z, tiles, mine = self.engine, run.z, run.tiles, run.mine
        N, dev = z.shape[0], z.device
        use_rec = REC_PATH and hasattr(E, "rec_from_f32")
        if use_rec:
            norm_ord = {i: k for k, i in enumerate(i for i, s in enumerate(steps) if s.kind == "norm")}
            coefs = [E.gn_coeffs(mean, var, steps[i].norm[0], steps[i].norm[1], steps[i].channels, 32, 1e-6)
                     for i, (var, mean) in zip(norm_ord, frozen)]
            for i in mine:
                run.live[i] = self._live_plan(steps, run.in_bboxes[i], run.out_bboxes[i])
        if use_rec and TILE_BATCH > 1:
            # Tiles of one shape go through the sweep TOGETHER (stacked along the batch axis, TILE_BATCH at a time).  Upstream
            # walks them one by one (:578-642); with frozen statistics they are independent, so the result is the same -- but
            # a conv launch over one tile fills the 256 CUs in ceil(items / 256) rounds and the last round is mostly empty
            # (256 -> 256 at 1112^2: 4 900 items = 19.1 rounds, 4 % idle; 512 -> 512 at 278^2: 2.5 rounds, 16 % idle).
            # 288 GB of HBM hold several tiles' activations at once (4 tiles of 278^2: ~53 GB).
            # Tiles stack by (shape, window SIZES of their narrowed upsample convs): each tile keeps its own window ORIGIN
            # (mdtile_upconv2d_rec_window takes one per image), so e.g. the interior tiles of every row share their launches.  A chunk
            # runs start to finish in ONE pass.
            rep_cache: Dict[int, tuple] = {1: (frozen, coefs)}

            def rep(T):                          # frozen statistics / coefficient rows of a T-deep stack (built once per depth)
                if T not in rep_cache:
                    rep_cache[T] = ([(v.repeat(T), m.repeat(T)) for v, m in frozen], [c.repeat(T, 1, 1) for c in coefs])
                return rep_cache[T]

            def regather(chunk):                 # inputs that were folded into a stacked copy: cut them out of z again
                for i in chunk:                  # (tiles of the chunk that already FINISHED -- an OOM inside finish() -- are None: left alone)
                    if tiles[i] is not None:
                        b = run.in_bboxes[i]
                        tiles[i].x = E.gather_rect(z, b[0], b[2], b[1] - b[0], b[3] - b[2])

            def run_stack(chunk):
                T = len(chunk)
                xb = tiles[chunk[0]].x if T == 1 else torch.cat([tiles[i].x for i in chunk], dim=0)     # (4-channel latent tiles: KBs)
                fz, cf = rep(T)
                if T > 1:
                    for i in chunk:
                        tiles[i].x = None          # the stacked copy is the live one
                w0 = run.live[chunk[0]][0]
                wins = {k: ([run.live[i][0][k][0] for i in chunk for _ in range(N)], [run.live[i][0][k][1] for i in chunk for _ in range(N)], w[2], w[3])
                        for k, w in w0.items()} if w0 else None
                yb = self._run_tile_rec(steps, xb, fz, cf, norm_ord, wins)
                for t, i in enumerate(chunk):
                    tiles[i].x = yb[t * N:(t + 1) * N]
                    run.finish(i)

            groups: Dict[tuple, List[int]] = {}
            for i in mine:
                groups.setdefault(tuple(tiles[i].x.shape[2:]) + tuple((k, w[2], w[3]) for k, w in sorted(run.live[i][0].items())), []).append(i)
            for key in sorted(groups, key=lambda kk: -len(groups[kk])):
                ids = groups[key]
                tb = self._tile_batch_that_fits(N, key[:2], dev)
                if len(key) > 2 and tb * N > 8:       # the group carries narrowed windows (key = shape + window entries):
                    tb = max(1, 8 // N)               # mdtile_upconv2d_rec_window keeps 8 window origins per launch
                c0 = 0
                while c0 < len(ids):
                    if state.interrupted:
                        run.interrupted = True
                        return
                    chunk = [i for i in ids[c0:c0 + tb] if tiles[i] is not None]      # (after an OOM retry: not the tiles that finished)
                    if not chunk:
                        c0 += tb
                        continue
                    try:
                        run_stack(chunk)
                    except torch.cuda.OutOfMemoryError:
                        if len(chunk) == 1:
                            raise
                        print(f"[Tiled VAE]: {len(chunk)} stacked tiles do not fit in VRAM, continuing one tile per sweep")
                        torch.cuda.empty_cache()
                        regather(chunk)
                        tb = 1
                        continue
                    c0 += tb
            return
        for i in mine:
            if state.interrupted:
                run.interrupted = True
                return
            st, k = tiles[i], 0
            if use_rec:
                st.x = self._run_tile_rec(steps, st.x, frozen, coefs, norm_ord, run.live[i][0])
                run.finish(i)
                continue
            while True:
                self._run_until_norm(steps, st)
                if st.pc >= len(steps):
                    break
                self._apply_norm(steps, st, *frozen[k])
                k += 1
            run.finish(i)

    def _sweep_lockstep(self, run: "VAEHook._Run", steps: List[Step], frozen) -> None:
        """Slow mode: all tiles advance in lockstep from norm to norm, the statistics pooled over the tiles (and ranks) at each one
        (upstream :289-361, :578-642 zig-zag); semi-fast (color_fix): the first len(frozen) norms use the frozen statistics instead."""
        tiles, mine = run.tiles, run.mine
        world = self.shard[1]
        forward, k_norm = True, 0
        while True:
            use_frozen = frozen is not None and k_norm < len(frozen)
            gp = GroupNormParam(self.engine)
            for i in (() if run.interrupted else mine if forward else reversed(mine)):
                if state.interrupted:
                    run.interrupted = True
                    break
                self._run_until_norm(steps, tiles[i], want_stats=not use_frozen)
                if tiles[i].pc < len(steps) and not use_frozen:
                    gp.add_tile(tiles[i].x, tiles[i].stats)
            if run.interrupted and world == 1:
                return
            # several ranks: an interrupted rank runs no more tiles but keeps walking the norms to the next POOLED barrier, where the
            # `head` exchange tells every rank (see _pooled_across_ranks) -- all of them leave this loop at the same barrier
            if use_frozen:
                # a frozen norm is no barrier upstream (the tile runs straight through it): no pooling, no collective,
                # no change of the zig-zag direction.  A later pooled norm always exists in this branch.
                for i in (() if run.interrupted else mine):
                    self._apply_norm(steps, tiles[i], *frozen[k_norm])
                k_norm += 1
                continue
            if world == 1:
                pooled = gp.summary()
            else:
                pooled, any_interrupted = self._pooled_across_ranks(gp, steps, run.z.device, run.interrupted)
                if any_interrupted:
                    run.interrupted = True
                    return
            k_norm += 1
            if pooled is None:
                for i in mine:
                    run.finish(i)
                return
            for i in mine:
                self._apply_norm(steps, tiles[i], *pooled)
            forward = not forward

    @torch.no_grad()
    def vae_tile_forward(self, z: Tensor) -> Tensor:
        t0 = time()
        net = self.net
        dev = next(net.parameters()).device
        dtype = next(net.parameters()).dtype
        E = self.engine
        E.require_device(dev)      # the HIP engine: raises unless the VAE sits on the GPU (no CPU path exists)
        z = z.detach().to(device=dev, dtype=torch.float32).contiguous()
        N, _, height, width = z.shape
        net.last_z_shape = z.shape
        print(f"[Tiled VAE]: input_size: {z.shape}, tile_size: {self.tile_size}, padding: {self.pad}")
        in_bboxes, out_bboxes = self.split_tiles(height, width)
        steps = self.program()
        rank, world = self.shard

        frozen = None
        if self.fast_mode:
            zs = E.vae_fast_input(z, self.tile_size)
            print(f"[Tiled VAE]: Fast mode enabled, estimating group norm parameters on {zs.shape[3]} x {zs.shape[2]} image")
            if world > 1 and SP_ESTIMATOR and self.is_decoder and zs.shape[2] >= 2 * world:
                # the estimator is one untiled pass: split it by rows across the ranks instead of repeating it on each
                from mdtile import seqpar
                frozen = seqpar.estimate_group_norm_sp(steps, zs, seqpar.BandComm(rank, world), self._sp_ops or seqpar.EngineOps(), FUSE_PRE_GN)
            else:
                frozen = self.estimate_group_norm(zs, steps)
        all_frozen = frozen is not None and len(frozen) == sum(1 for s in steps if s.kind == "norm")
        if all_frozen and self.devices and len(self.devices) > 1 and dev.type == "cuda":
            return self._multi_device_sweep(z, steps, frozen, in_bboxes, out_bboxes, dtype, t0)

        owner = [0] * len(in_bboxes)
        if world > 1:
            from mdtile import sharding as _sh
            owner = _sh.deal_tiles(in_bboxes, world)      # by tile area (mdtile/sharding.py: deal_tiles), the same list on every rank
        mine = [i for i in range(len(in_bboxes)) if owner[i] == rank] if world > 1 else list(range(len(in_bboxes)))
        run = VAEHook._Run(self, z, in_bboxes, out_bboxes, mine)
        if all_frozen:
            self._sweep_frozen(run, steps, frozen)
        else:
            self._sweep_lockstep(run, steps, frozen)
        return self._assemble(run, owner, dtype, t0)

    def _assemble(self, run: "VAEHook._Run", owner, dtype, t0) -> Tensor:
        """NaN test of the image (upstream :633-634), the gather of the other ranks' rectangles, upstream's interrupt results (:644-650)."""
        net, z, result, interrupted = self.net, run.z, run.result, run.interrupted
        dev = z.device
        N, _, height, width = z.shape
        rank, world = self.shard
        nan_seen = bool(run.nan_flags) and bool(torch.stack(run.nan_flags).any().item())
        if world > 1 and self.gather_to is not None:
            # The gather below is a grouped exchange EVERY rank must enter (or none): a rank that was interrupted, or whose NaN check
            # raises, would leave the others -- and the root's receives -- waiting for ever.  So the ranks first agree on both flags (one
            # small all-reduce), then skip the gather together / raise together.  Upstream tests every tile of the image it returns
            # (tilevae.py:633-634): with the flags summed the root sees a NaN found on any rank.
            from mdtile import sharding
            agree = torch.tensor([1.0 if interrupted else 0.0, 1.0 if nan_seen else 0.0], dtype=torch.float32, device=dev)
            sharding.comm_allreduce_sum(agree)
            flags = agree.tolist()
            interrupted, nan_seen = flags[0] > 0.0, flags[1] > 0.0
        if nan_seen:
            devices.test_for_nans(torch.full((1,), float("nan")), "vae")     # raises the host's NansException (or not: --disable-nan-check)
        if world > 1 and self.gather_to is not None and not interrupted:
            from mdtile import sharding
            if result is None and rank == self.gather_to:
                result = torch.zeros((N, 3 if self.is_decoder else 2 * int(getattr(net, "z_channels", 4)),
                                      *((height * 8, width * 8) if self.is_decoder else (height // 8, width // 8))), device=dev, dtype=torch.float32)
            if result is not None:
                sharding.gather_tiles_to_root(result, run.out_bboxes, lambda i: owner[i], rank, self.gather_to)
        self.last_seconds = time() - t0
        if interrupted and result is not None:
            return result.to(dtype)          # upstream hands back what is finished (:644-647)
        if result is None:
            if not self.is_decoder:
                raise RuntimeError("[Tiled VAE]: interrupted before any encoder tile finished")
            from modules.sd_vae_approx import cheap_approximation
            approx = torch.cat([torch.nn.functional.interpolate(cheap_approximation(x).unsqueeze(0), scale_factor=8,
                                                                mode="nearest-exact") for x in z], dim=0)
            return approx.to(dev, dtype=dtype)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
            print(f"[Tiled VAE]: Done in {time() - t0:.3f}s, max VRAM alloc {torch.cuda.max_memory_allocated(dev) / 2**20:.3f} MB")
        return result.to(dtype)


class Script(scripts.Script):

    def __init__(self):
        self.hooked = False

    def title(self):
        return "Tiled VAE"

    def show(self, is_img2img):
        return scripts.AlwaysVisible

    def ui(self, is_img2img):
        import gradio as gr
        tab = "t2i" if not is_img2img else "i2i"
        uid = lambda name: f"MD-{tab}-{name}"  # noqa: E731
        with gr.Accordion("Tiled VAE", open=False, elem_id=f"MDV-{tab}"):
            with gr.Row():
                enabled = gr.Checkbox(label="Enable Tiled VAE", value=False, elem_id=uid("enable"))
                vae_to_gpu = gr.Checkbox(label="Move VAE to GPU (if possible)", value=True, elem_id=uid("vae2gpu"))
            with gr.Row():
                encoder_tile_size = gr.Slider(label="Encoder Tile Size", minimum=256, maximum=4096, step=16,
                                              value=get_rcmd_enc_tsize(), elem_id=uid("enc-size"))
                decoder_tile_size = gr.Slider(label="Decoder Tile Size", minimum=48, maximum=512, step=16,
                                              value=get_rcmd_dec_tsize(), elem_id=uid("dec-size"))
            with gr.Row():
                fast_encoder = gr.Checkbox(label="Fast Encoder", value=True, elem_id=uid("fastenc"))
                color_fix = gr.Checkbox(label="Fast Encoder Color Fix", value=False, elem_id=uid("fastenc-colorfix"))
                fast_decoder = gr.Checkbox(label="Fast Decoder", value=True, elem_id=uid("fastdec"))
        return [enabled, encoder_tile_size, decoder_tile_size, vae_to_gpu, fast_decoder, fast_encoder, color_fix]

    def process(self, p, enabled: bool, encoder_tile_size: int, decoder_tile_size: int, vae_to_gpu: bool,
                fast_decoder: bool, fast_encoder: bool, color_fix: bool):
        vae = p.sd_model.first_stage_model
        decoder = vae.decoder
        if not enabled:
            if self.hooked:
                for net in (decoder, getattr(vae, "encoder", None)):
                    if net is not None and isinstance(net.forward, VAEHook):
                        net.forward.net = None
                        net.forward = net.original_forward
            self.hooked = False
            return
        if not hasattr(decoder, "original_forward"):
            decoder.original_forward = decoder.forward
        self.hooked = True
        decoder.forward = VAEHook(decoder, decoder_tile_size, is_decoder=True, fast_decoder=fast_decoder,
                                  fast_encoder=fast_encoder, color_fix=color_fix, to_gpu=vae_to_gpu)
        encoder = vae.encoder
        if not hasattr(encoder, "original_forward"):
            encoder.original_forward = encoder.forward
        encoder.forward = VAEHook(encoder, encoder_tile_size, is_decoder=False, fast_decoder=fast_decoder,
                                  fast_encoder=fast_encoder, color_fix=color_fix, to_gpu=vae_to_gpu)

    def postprocess(self, p, processed, enabled: bool, *args):
        if not enabled:
            return
        vae = p.sd_model.first_stage_model
        for net in (vae.decoder, getattr(vae, "encoder", None)):
            if net is not None and isinstance(net.forward, VAEHook):
                net.forward.net = None
                net.forward = net.original_forward
