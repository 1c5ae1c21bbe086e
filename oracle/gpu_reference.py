"""
ORACLE (test infrastructure only) -- the SAME oracle code (oracle/vae_oracle.py, pinned bit-exact to upstream) executed with cuda
tensors: torch's native fp32 conv (im2col + rocBLAS sgemm) and bmm / softmax are the arithmetic engine, i.e. an implementation
independent of libmdtile.so, for the tile sizes the CPU oracle cannot finish in minutes (decoder tile 256: 278x278-latent tiles,
77 284-token attention, 2224x2224 convs).  Only tests/ and bench.py's `parity` leg (untimed) may import this file.

Two adaptations, both exact:
  * big stride-1 'same' convs are evaluated in horizontal bands with a one-row halo (torch's im2col index is 32-bit; rows of a conv
    are independent); likewise the encoder's stride-2 Downsample conv, in bands of output rows;
  * the T x T attention matrix of tile_utils/attn.py:49-72 is formed for `chunk` queries at a time (23.9 GB otherwise; the softmax
    is row-wise).
MIOpen is switched off while the reference runs: a fresh box has no kernel cache and every new conv shape would JIT for tens of seconds.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F

from oracle import vae_oracle as vo

_orig_conv2d = F.conv2d


def banded_conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    k = w.shape[-1]
    same = stride in (1, (1, 1)) and padding in (k // 2, (k // 2, k // 2)) and dilation in (1, (1, 1)) and groups == 1
    cols = x.shape[1] * k * k * x.shape[2] * x.shape[3]
    down = stride in (2, (2, 2)) and padding in (0, (0, 0)) and dilation in (1, (1, 1)) and groups == 1 and k == 3
    if x.is_cuda and down and (cols // 4 > 2 ** 29 or w.shape[0] * x.shape[2] * x.shape[3] // 4 > 2 ** 29):
        # ldm Downsample (encoder, scripts/tilevae.py:155-171): 3x3 stride 2 over an explicitly padded input -- output row r reads input rows
        # 2r .. 2r+2, so bands of output rows are independent too (same 32-bit im2col limit)
        Ho = (x.shape[2] - 3) // 2 + 1
        band = max(8, min((2 ** 28) // (x.shape[1] * k * k * ((x.shape[3] - 3) // 2 + 1)), (2 ** 28) // (w.shape[0] * ((x.shape[3] - 3) // 2 + 1))))
        outs = []
        for y0 in range(0, Ho, band):
            y1 = min(Ho, y0 + band)
            outs.append(_orig_conv2d(x[:, :, 2 * y0:2 * (y1 - 1) + 3], w, b, 2, 0, 1, 1))
        return torch.cat(outs, dim=2)
    # (the OUTPUT counts too: torch's native conv writes rows past 2^32 bytes of its result to the wrong place -- conv_in 3 -> 128 on a
    # 3072 x 3072 encoder image has a small im2col buffer and a 4.8 GB result; probes/enc_debug.py, profiles/r5k)
    outn = w.shape[0] * x.shape[2] * x.shape[3]
    if not (x.is_cuda and same and (cols > 2 ** 29 or outn > 2 ** 29)):
        return _orig_conv2d(x, w, b, stride, padding, dilation, groups)
    H, h = x.shape[2], k // 2
    band = max(8, min((2 ** 28) // (x.shape[1] * k * k * x.shape[3]), (2 ** 28) // (w.shape[0] * x.shape[3])))
    outs = []
    for y0 in range(0, H, band):
        y1 = min(H, y0 + band)
        lo, hi = max(0, y0 - h), min(H, y1 + h)
        xb = F.pad(x[:, :, lo:hi], (h, h, h - (y0 - lo), h - (hi - y1)))
        outs.append(_orig_conv2d(xb, w, b, 1, 0, 1, 1))
    return torch.cat(outs, dim=2)


def attn_body_chunked(attn, h_, chunk=4096):
    """vo.attn_body (tile_utils/attn.py:49-72) with the queries processed `chunk` at a time."""
    q, k, v = attn.q(h_), attn.k(h_), attn.v(h_)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    v = v.reshape(b, c, hh * ww)
    out = torch.empty_like(v)
    for i in range(0, hh * ww, chunk):
        w_ = torch.softmax(torch.bmm(q[:, i:i + chunk], k) * (int(c) ** (-0.5)), dim=2)
        out[:, :, i:i + chunk] = torch.bmm(v, w_.permute(0, 2, 1))
    return attn.proj_out(out.reshape(b, c, hh, ww))


@contextlib.contextmanager
def reference_arithmetic(chunked_attention: bool = True):
    old_conv, old_attn = F.conv2d, vo.attn_body
    F.conv2d = banded_conv2d
    if chunked_attention:
        vo.attn_body = attn_body_chunked
    try:
        with torch.backends.cudnn.flags(enabled=False):
            yield
    finally:
        F.conv2d, vo.attn_body = old_conv, old_attn


@torch.no_grad()
def tiled_forward_gpu(net, z: torch.Tensor, tile_size: int, fast: bool, is_decoder: bool = True, color_fix: bool = False, only_tiles=None):
    """vo.tiled_forward (upstream vae_tile_forward, scripts/tilevae.py:507-656) on the device `net` lives on.
    only_tiles: see vo.tiled_forward -- [(out_bbox, cropped tile), ...] of the listed tiles (fast mode)."""
    dev = next(net.parameters()).device
    with reference_arithmetic():
        return vo.tiled_forward(net, z.to(dev), tile_size, fast, is_decoder=is_decoder, color_fix=color_fix, only_tiles=only_tiles)
