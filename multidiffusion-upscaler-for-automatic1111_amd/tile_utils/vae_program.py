"""
The Tiled-VAE PROGRAM and its geometry helpers: upstream's task queue (scripts/tilevae.py:107-204) with the fusions the engine offers
already applied, the live-window arithmetic of the fast-mode decoder, crop_valid_region (:248-259), the slow-mode statistics collector
(GroupNormParam, :289-335) and the per-tile executor state.  Pure host code: no switches, no device work besides what the packed
convs / the engine do when called.  scripts/tilevae.py (the plugin surface: Script, VAEHook) re-exports every name defined here.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor

import mdtile


# ---------------------------------------------------------------------------------------------------------------------
# program = upstream's task queue, with the fusions the engine offers already applied
# ---------------------------------------------------------------------------------------------------------------------

class Step:
    __slots__ = ("kind", "conv", "norm", "silu", "attn", "fuse_res", "upsample", "downsample", "channels")

    def __init__(self, kind, conv=None, norm=None, silu=False, attn=None, fuse_res=False, upsample=False, downsample=False, channels=0):
        self.kind, self.conv, self.norm, self.silu, self.attn = kind, conv, norm, silu, attn
        self.fuse_res, self.upsample, self.downsample = fuse_res, upsample, downsample
        self.channels = channels      # norm steps: the GroupNorm's channel count


class AttnPack:
    """q/k/v/proj_out of one AttnBlock, packed for the engine; v is produced token-major for the PV contraction."""

    def __init__(self, attn, pack=None, engine=None):
        pack = pack or _pack
        self.engine = engine or mdtile
        self.q, self.k, self.v, self.proj = (pack(attn.q), pack(attn.k), pack(attn.v), pack(attn.proj_out))
        self.channels = attn.q.weight.shape[0]

    def __call__(self, h: Tensor, residual: Tensor) -> Tensor:
        B, C, H, W = h.shape
        q = self.q(h).view(B, C, H * W)
        k = self.k(h).view(B, C, H * W)
        scale = float(int(C) ** (-0.5))
        if getattr(self.engine, "v_channel_major_ok", lambda c: False)(C):
            # v like q and k: channel-major through the split-bf16 1x1 kernel; the attention prep reads it in that layout
            o = self.engine.vae_attn(q, k, self.v(h).view(B, C, H * W), scale, v_channel_major=True)
        else:
            o = self.engine.vae_attn(q, k, self.v(h, token_major=True), scale)     # softmax(q^T k / sqrt(C)) v   (attn.py:55-67)
        return self.proj(o.view(B, C, H, W), residual=residual)        # proj_out + the queue's add_res


def _pack(conv) -> mdtile.PackedConv:
    if conv.stride == (2, 2):
        # ldm Downsample.conv: 3x3, stride 2, no padding (the module pads right/bottom by one itself) -> PackedConv.down2
        assert conv.kernel_size == (3, 3) and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1, f"unsupported conv {conv}"
        return mdtile.PackedConv(conv.weight.detach().float().contiguous(), None if conv.bias is None else conv.bias.detach().float())
    assert conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1, "engine convs are stride-1 (or ldm Downsample) dense"
    k = conv.kernel_size[0]
    assert conv.kernel_size == (k, k) and conv.padding == (k // 2, k // 2), f"unsupported conv {conv}"
    return mdtile.PackedConv(conv.weight.detach().float().contiguous(),
                             None if conv.bias is None else conv.bias.detach().float())


def _norm_params(gn):
    assert gn.num_groups == 32, "Tiled VAE hard-codes 32 groups (upstream tilevae.py:299)"
    g = gn.weight.detach().float().contiguous() if getattr(gn, "weight", None) is not None else None
    b = gn.bias.detach().float().contiguous() if getattr(gn, "bias", None) is not None else None
    return g, b


def _resblock(steps: List[Step], blk, pack):
    if blk.in_channels != blk.out_channels:
        shortcut = blk.conv_shortcut if blk.use_conv_shortcut else blk.nin_shortcut
        steps.append(Step("store_res", conv=pack(shortcut)))
    else:
        steps.append(Step("store_res"))
    steps.append(Step("norm", norm=_norm_params(blk.norm1), silu=True, channels=blk.norm1.num_channels))
    steps.append(Step("conv", conv=pack(blk.conv1)))
    steps.append(Step("norm", norm=_norm_params(blk.norm2), silu=True, channels=blk.norm2.num_channels))
    steps.append(Step("conv", conv=pack(blk.conv2), fuse_res=True))       # conv2 + add_res in one epilogue


def build_task_queue(net, is_decoder: bool = True, pack=None, engine=None) -> List[Step]:
    """Linearise an ldm Decoder exactly in upstream's order (:139-195): conv_in, mid(res, attn, res), levels top-down
    with num_res_blocks+1 resblocks (+ upsample except on level 0), norm_out, silu, conv_out.  30 norms for SD/SDXL.
    `pack` turns an nn.Conv2d into the callable a step carries and `engine` is the module the attention step calls
    (defaults: mdtile.PackedConv / mdtile; the CPU tests of the host logic inject torch doubles, tests/torch_engine.py)."""
    pack = pack or _pack
    steps = [Step("conv", conv=pack(net.conv_in))]

    def _mid():
        _resblock(steps, net.mid.block_1, pack)
        steps.extend([Step("store_res"), Step("norm", norm=_norm_params(net.mid.attn_1.norm), channels=net.mid.attn_1.norm.num_channels),
                      Step("attn", attn=AttnPack(net.mid.attn_1, pack, engine))])
        _resblock(steps, net.mid.block_2, pack)

    if is_decoder:
        _mid()
        for lvl in reversed(range(net.num_resolutions)):
            for i in range(net.num_res_blocks + 1):
                _resblock(steps, net.up[lvl].block[i], pack)
            if lvl != 0:
                steps.append(Step("conv", conv=pack(net.up[lvl].upsample.conv), upsample=True))  # nearest-2x fused
    else:
        # encoder (upstream :155-171): levels bottom-up with num_res_blocks resblocks (+ downsample except on the last), then mid
        for lvl in range(net.num_resolutions):
            for i in range(net.num_res_blocks):
                _resblock(steps, net.down[lvl].block[i], pack)
            if lvl != net.num_resolutions - 1:
                steps.append(Step("conv", conv=pack(net.down[lvl].downsample.conv), downsample=True))
        _mid()
    if not is_decoder or not net.give_pre_end:
        steps.append(Step("norm", norm=_norm_params(net.norm_out), silu=True, channels=net.norm_out.num_channels))
        steps.append(Step("conv", conv=pack(net.conv_out)))
        if is_decoder and net.tanh_out:
            steps.append(Step("tanh"))
    return steps



def crop_valid_region(x, input_bbox, target_bbox, is_decoder):
    padded = [i * 8 if is_decoder else i // 8 for i in input_bbox]
    m = [target_bbox[i] - padded[i] for i in range(4)]
    return x[:, :, m[2]:x.size(2) + m[3], m[0]:x.size(3) + m[1]]


def live_windows(steps: List["Step"], tile_hw: Tuple[int, int], valid: Tuple[int, int, int, int]):
    """Live-window narrowing of ONE decoder tile whose GroupNorm statistics are all frozen (fast mode).

    Upstream decodes the whole padded tile and crop_valid_region (:248-259, applied at :630-632) keeps `valid` (y0, x0, y1, x1 in latent
    px relative to the tile; the padding -- 11 latent px for the decoder, :371 -- is thrown away).  With frozen statistics every layer
    behind the attention is local (3x3 convs, 1x1 convs, pointwise norm / SiLU, nearest 2x), so an output pixel further than the number
    of 3x3 convs still to come from the valid region cannot reach it and need not be computed.  The plane is narrowed where it is
    cheapest, at the upsample convs: walking the program backwards, `need` counts the 3x3 convs behind a point in pixels of that
    level; at the upsample conv that opens a level of `scale` px per latent px the plane becomes `valid` grown by
    ceil(need / scale) latent px (whole latent px: the tile's own crop and store stay in latent units), clamped to the tile, and
    the level below has to provide ceil((need + 1) / 2) px: an output pixel d px outside the valid region reads the nearest-2x
    image d - 1 .. d + 1 px outside, i.e. input pixels up to ceil((d + 1) / 2) px outside (the window's own outermost inputs come
    from the un-narrowed input image, so they are its true neighbours).  A narrowed plane is a zero-padded image of its own: its
    errors creep inwards one pixel per conv and stop at the valid region.  The walk ends at the attention (it needs every token).
    SD decoder (3 resblocks per level, conv_out): need = 7, 10, 12 px -> grow = 1, 3, 6 latent px for the 8x, 4x, 2x levels; the 1x
    level would need 13 of its 11 px of padding and stays whole.  (tests/test_vae_host_logic.py: exact and tight in float64.)

    Returns ({index of the upsample step: (y0, x0, h, w) window of ITS input plane, in input px}, final rect in latent px relative to
    the tile (y0, x0, y1, x1)) -- ({}, whole tile) when nothing can be shed."""
    th, tw = tile_hw
    whole = (0, 0, th, tw)
    ups = [i for i, s in enumerate(steps) if s.kind == "conv" and s.upsample]
    if not ups or any(s.kind == "conv" and s.downsample for s in steps):
        return {}, whole
    scale, need, grow = 1 << len(ups), 0, {}
    for i in range(len(steps) - 1, -1, -1):
        s = steps[i]
        if s.kind == "attn":
            break
        if s.kind != "conv":
            continue                       # frozen norm, SiLU, residual bookkeeping (+ 1x1 nin_shortcut), tanh: pointwise
        ks = int(getattr(s.conv, "ksize", 3))
        if s.upsample:
            grow[i] = -(-need // scale)    # whole latent px
            scale //= 2
            need = (need + ks // 2 + 1) // 2
        else:
            need += ks // 2
    vy0, vx0, vy1, vx1 = valid
    windows, cur, in_scale = {}, whole, 1
    for i in ups:
        if i in grow:
            m = grow[i]
            rect = (max(cur[0], vy0 - m), max(cur[1], vx0 - m), min(cur[2], vy1 + m), min(cur[3], vx1 + m))
            if rect != cur:
                windows[i] = ((rect[0] - cur[0]) * in_scale, (rect[1] - cur[1]) * in_scale,
                              (rect[2] - rect[0]) * in_scale, (rect[3] - rect[1]) * in_scale)
                cur = rect
        in_scale *= 2
    return windows, cur


class GroupNormParam:
    """Slow-mode collector: per-tile (var, mean) rows pooled by pixel count (upstream :289-335)."""

    def __init__(self, engine=None):
        self.engine = engine or mdtile
        self.var_list, self.mean_list, self.pixel_list = [], [], []

    def add_tile(self, tile: Tensor, stats=None):
        """stats: (var, mean) of `tile` when its producer has already left them (TileState.stats); else one pass over the tile."""
        var, mean = stats if stats is not None else self.engine.gn_stats(tile, 32)
        self.var_list.append(var)
        self.mean_list.append(mean)
        self.pixel_list.append(tile.shape[2] * tile.shape[3])

    def summary(self) -> Optional[Tuple[Tensor, Tensor]]:
        if not self.var_list:
            return None
        return self.engine.gn_pool(torch.vstack(self.mean_list), torch.vstack(self.var_list), self.pixel_list)


# ---------------------------------------------------------------------------------------------------------------------
class TileState:
    __slots__ = ("x", "res", "pc", "pre", "stats")

    def __init__(self, x):
        self.x, self.res, self.pc = x, [], 0
        self.pre = None   # pending fused pre-activation: gn_coeffs of the norm just resolved, consumed by the next conv
        self.stats = None  # slow mode: (var, mean) of x, left by the conv that produced it (None: nobody has them yet)
