"""
ORACLE (test infrastructure only) -- CPU restatement of the reference's Tiled-VAE arithmetic (scripts/tilevae.py).

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this file.

Parity status: PINNED by tests/test_oracle_vs_reference.py (runs upstream `VAEHook` itself under
hostsim/stub_host.py when /root/reference is mounted) and by tests/golden/vae_*.npz (produced by the upstream code,
see tests/golden/make_golden.py).

The restatement is functional: the decoder is linearised into a flat op list (what upstream calls the task
queue, tilevae.py:107-195), and the executor replays upstream's GroupNorm barrier semantics
(tilevae.py:507-656) without its CPU<->GPU tile parking, which has no numerical effect.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

DEC_PAD, ENC_PAD = 11, 32          # tilevae.py:373


# --------------------------------------------------------------------------------------------------
# tile geometry  (tilevae.py:390-462, 248-259)
# --------------------------------------------------------------------------------------------------
def best_tile_size(lower: int, upper: int) -> int:
    """tilevae.py:390-403 -- round `lower` up to a multiple of 32/16/8/4/2 (first that fits under `upper`)."""
    div = 32
    while div >= 2:
        rem = lower % div
        if rem == 0:
            return lower
        cand = lower - rem + div
        if cand <= upper:
            return cand
        div //= 2
    return lower


def split_tiles(h: int, w: int, tile_size: int, is_decoder: bool = True):
    """tilevae.py:405-462.  bbox order is [x1, x2, y1, y2]."""
    pad = DEC_PAD if is_decoder else ENC_PAD
    nh = max(math.ceil((h - 2 * pad) / tile_size), 1)
    nw = max(math.ceil((w - 2 * pad) / tile_size), 1)
    rh = best_tile_size(math.ceil((h - 2 * pad) / nh), tile_size)
    rw = best_tile_size(math.ceil((w - 2 * pad) / nw), tile_size)
    ins, outs = [], []
    for i in range(nh):
        for j in range(nw):
            box = [pad + j * rw, min(pad + (j + 1) * rw, w), pad + i * rh, min(pad + (i + 1) * rh, h)]
            out = [box[0] if box[0] > pad else 0, box[1] if box[1] < w - pad else w,
                   box[2] if box[2] > pad else 0, box[3] if box[3] < h - pad else h]
            outs.append([v * 8 if is_decoder else v // 8 for v in out])
            ins.append([max(0, box[0] - pad), min(w, box[1] + pad), max(0, box[2] - pad), min(h, box[3] + pad)])
    return ins, outs


def crop_valid_region(x: torch.Tensor, in_bbox, out_bbox, is_decoder: bool = True) -> torch.Tensor:
    """tilevae.py:248-259."""
    padded = [v * 8 if is_decoder else v // 8 for v in in_bbox]
    m = [out_bbox[i] - padded[i] for i in range(4)]
    return x[:, :, m[2]:x.size(2) + m[3], m[0]:x.size(3) + m[1]]


# --------------------------------------------------------------------------------------------------
# GroupNorm with frozen statistics  (tilevae.py:207-245, 289-361)
# --------------------------------------------------------------------------------------------------
def get_var_mean(x: torch.Tensor, groups: int = 32):
    """tilevae.py:207-215 -- biased variance and mean per (sample, group)."""
    b, c = x.shape[:2]
    v = x.contiguous().view(1, b * groups, c // groups, *x.shape[2:])
    return torch.var_mean(v, dim=[0, 2, 3, 4], unbiased=False)


def custom_group_norm(x, groups, mean, var, weight=None, bias=None, eps: float = 1e-6):
    """tilevae.py:218-245 -- (x-mean)/sqrt(var+eps) per (sample, group), then per-channel affine."""
    b, c = x.shape[:2]
    v = x.contiguous().view(1, b * groups, c // groups, *x.shape[2:])
    out = F.batch_norm(v, mean.to(x), var.to(x), None, None, training=False, momentum=0, eps=eps).view(b, c, *x.shape[2:])
    if weight is not None:
        out = out * weight.view(1, -1, 1, 1)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def pool_stats(vars_: Sequence[torch.Tensor], means: Sequence[torch.Tensor], pixels: Sequence[int]):
    """tilevae.py:320-335 -- pixel-count weighted mean of per-tile means AND of per-tile variances (the between-tile
    spread of the means is ignored upstream; do not 'fix')."""
    px = torch.tensor(list(pixels), dtype=torch.float32) / max(pixels)
    p = (px / px.sum()).unsqueeze(1).to(vars_[0].device)      # (formed on the host like upstream; the device only when run via gpu_reference)
    return (torch.vstack(list(vars_)) * p).sum(0), (torch.vstack(list(means)) * p).sum(0)


# --------------------------------------------------------------------------------------------------
# attention body without norm / residual  (tile_utils/attn.py:49-72)
# --------------------------------------------------------------------------------------------------
def attn_body(attn, h_: torch.Tensor) -> torch.Tensor:
    q, k, v = attn.q(h_), attn.k(h_), attn.v(h_)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
    v = v.reshape(b, c, hh * ww)
    out = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return attn.proj_out(out)


# --------------------------------------------------------------------------------------------------
# op list  (tilevae.py:107-195)
# --------------------------------------------------------------------------------------------------
def _res_ops(ops: list, blk):
    short = blk.nin_shortcut if blk.in_channels != blk.out_channels and not blk.use_conv_shortcut else \
        (blk.conv_shortcut if blk.in_channels != blk.out_channels else None)
    ops.append(("store_res", short))
    ops += [("norm", blk.norm1), ("silu", None), ("conv", blk.conv1),
            ("norm", blk.norm2), ("silu", None), ("conv", blk.conv2), ("add_res", None)]


def _attn_ops(ops: list, attn):
    ops += [("store_res", None), ("norm", attn.norm), ("attn", attn), ("add_res", None)]


def build_ops(net, is_decoder: bool = True) -> list:
    ops = [("conv", net.conv_in)]
    if is_decoder:
        _res_ops(ops, net.mid.block_1)
        _attn_ops(ops, net.mid.attn_1)
        _res_ops(ops, net.mid.block_2)
        for lvl in reversed(range(net.num_resolutions)):
            for i in range(net.num_res_blocks + 1):
                _res_ops(ops, net.up[lvl].block[i])
            if lvl != 0:
                ops.append(("resample", net.up[lvl].upsample))
    else:
        for lvl in range(net.num_resolutions):
            for i in range(net.num_res_blocks):
                _res_ops(ops, net.down[lvl].block[i])
            if lvl != net.num_resolutions - 1:
                ops.append(("resample", net.down[lvl].downsample))
        _res_ops(ops, net.mid.block_1)
        _attn_ops(ops, net.mid.attn_1)
        _res_ops(ops, net.mid.block_2)
    if not is_decoder or not net.give_pre_end:
        ops += [("norm", net.norm_out), ("silu", None), ("conv", net.conv_out)]
        if is_decoder and net.tanh_out:
            ops.append(("tanh", None))
    return ops


def _run_segment(ops, start: int, tile: torch.Tensor, res_stack: list, norm_fn: Callable, stop_at_resample: bool = False):
    """Run ops[start:] on one tile until the next 'norm' (exclusive) or the end.  Returns (tile, next_index).
    stop_at_resample: also stop (index of the op, not executed) at the first 'resample' -- the encoder's color_fix estimate."""
    i = start
    while i < len(ops):
        kind, mod = ops[i]
        if kind == "norm" or (stop_at_resample and kind == "resample"):
            return tile, i
        if kind == "store_res":
            res_stack.append(tile if mod is None else mod(tile))
        elif kind == "add_res":
            tile = tile + res_stack.pop()
        elif kind == "silu":
            tile = F.silu(tile)
        elif kind == "attn":
            tile = attn_body(mod, tile)
        elif kind == "tanh":
            tile = torch.tanh(tile)
        else:  # conv / resample
            tile = mod(tile)
        i += 1
    return tile, i


def fast_mode_input(z: torch.Tensor, tile_size: int) -> torch.Tensor:
    """tilevae.py:545-559 -- nearest-exact downsample to <= tile_size, per-channel re-standardisation with the
    UNBIASED std of the full latent, clamp to the full latent's range."""
    h, w = z.shape[2:]
    sf = tile_size / max(h, w)
    d = F.interpolate(z, scale_factor=sf, mode="nearest-exact")
    std_old, mean_old = torch.std_mean(z, dim=[0, 2, 3], keepdim=True)
    std_new, mean_new = torch.std_mean(d, dim=[0, 2, 3], keepdim=True)
    d = (d - mean_new) / std_new * std_old + mean_old
    return torch.clamp(d, min=z.min(), max=z.max())


def estimate_stats(ops, z_small: torch.Tensor, color_fix: bool = False) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """tilevae.py:464-505 -- run the whole op list on the small latent, freezing (var, mean) at every norm.
    color_fix (encoder only, :492-496): the estimate returns at the first downsample; only the norms before it are frozen."""
    stats, res, tile, i = [], [], z_small, 0
    n_norm = sum(1 for k, _ in ops if k == "norm")
    while True:
        tile, i = _run_segment(ops, i, tile, res, None, stop_at_resample=color_fix)
        if i >= len(ops) or ops[i][0] == "resample":
            break
        var, mean = get_var_mean(tile, 32)
        stats.append((var, mean))
        if len(stats) == n_norm:          # upstream returns at the last norm without applying it
            break
        gn = ops[i][1]
        tile = custom_group_norm(tile, 32, mean, var, gn.weight, gn.bias)
        i += 1
    return stats


@torch.no_grad()
def tiled_forward(net, z: torch.Tensor, tile_size: int, fast: bool, is_decoder: bool = True, color_fix: bool = False,
                  only_tiles: Optional[Sequence[int]] = None):
    """tilevae.py:375-388 + 507-656.  Returns fp32 [N, C_out, 8H, 8W] (decoder).
    only_tiles (fast mode with EVERY norm frozen only): decode just these tiles of upstream's split (indices into split_tiles' list) and
    return [(out_bbox, cropped tile), ...] instead of the assembled image.  With all statistics frozen by the estimator
    (tilevae.py:464-505, 586-589) a tile's result does not depend on any other tile, so this is exactly what the full sweep writes into
    those rectangles -- the way to check single tiles of an image too large to decode whole on the checker (8K: 16 tiles)."""
    pad = DEC_PAD if is_decoder else ENC_PAD
    N, _, H, W = z.shape
    if max(H, W) <= pad * 2 + tile_size:                                   # tilevae.py:381-384
        assert only_tiles is None
        return net(z)
    ins, outs = split_tiles(H, W, tile_size, is_decoder)
    ops = build_ops(net, is_decoder)
    frozen = estimate_stats(ops, fast_mode_input(z, tile_size), color_fix and not is_decoder) if fast else None
    if only_tiles is not None:
        assert frozen is not None and len(frozen) == sum(1 for k, _ in ops if k == "norm"), "only_tiles needs every GroupNorm frozen (fast mode)"
        ins, outs = [ins[t] for t in only_tiles], [outs[t] for t in only_tiles]
    tiles = [z[:, :, b[2]:b[3], b[0]:b[1]].clone() for b in ins]
    T = len(tiles)
    pos = [0] * T
    res = [[] for _ in range(T)]
    norm_idx = 0
    result = None
    forward = True                       # upstream visits tiles zig-zag (tilevae.py:580-642); only the order of the
    while True:                          # per-tile stat rows (hence fp32 summation order) depends on it
        vars_, means, pixels = [], [], []
        for t in (range(T) if forward else reversed(range(T))):
            tiles[t], pos[t] = _run_segment(ops, pos[t], tiles[t], res[t], None)
            if pos[t] < len(ops):
                var, mean = get_var_mean(tiles[t], 32)
                vars_.append(var)
                means.append(mean)
                pixels.append(tiles[t].shape[2] * tiles[t].shape[3])
        if pos[0] >= len(ops):
            break
        is_frozen = frozen is not None and norm_idx < len(frozen)
        if is_frozen:
            var, mean = frozen[norm_idx]
        else:
            var, mean = pool_stats(vars_, means, pixels)
        gn = ops[pos[0]][1]
        for t in range(T):
            tiles[t] = custom_group_norm(tiles[t], 32, mean, var, gn.weight, gn.bias)
            pos[t] += 1
        norm_idx += 1
        if not is_frozen:                # a frozen norm is an inline task upstream: the sweep (and its direction) goes on
            forward = not forward
    if only_tiles is not None:
        return [(outs[t], crop_valid_region(tiles[t], ins[t], outs[t], is_decoder)) for t in range(T)]
    for t in range(T):
        tile = tiles[t]
        if result is None:                                                  # tilevae.py:629-632
            result = torch.zeros((N, tile.shape[1], H * 8 if is_decoder else H // 8, W * 8 if is_decoder else W // 8))
        ob = outs[t]
        result[:, :, ob[2]:ob[3], ob[0]:ob[1]] = crop_valid_region(tile, ins[t], ob, is_decoder).to(result.device)
    return result
