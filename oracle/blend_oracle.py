"""
ORACLE (test infrastructure only) -- CPU restatement of the reference's tile-blend arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this file.
The shipped path (multidiffusion-upscaler-for-automatic1111_amd/) never does.

Parity status: PINNED.  The reference ships no golden vectors, so this restatement is
pinned by (a) tests/test_oracle_vs_reference.py, which runs the upstream Python itself under
hostsim/stub_host.py whenever /root/reference is mounted, and (b) tests/golden/*.npz, which were produced
by the upstream code (tests/golden/make_golden.py) and travel to the GPU box.

Every function cites the upstream lines it follows.  All arithmetic is fp32 torch-on-CPU (grid ints are
Python ints, Gaussian profile is numpy float64 -> fp32, exactly as upstream).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

BG, FG = "Background", "Foreground"


# --------------------------------------------------------------------------------------------------
# grid  (tile_utils/utils.py:160-177, tile_methods/abstractdiffusion.py:173-186)
# --------------------------------------------------------------------------------------------------
def grid_origins(extent: int, tile: int, overlap: int) -> List[int]:
    """1-D tile origins: count = ceil((extent-ov)/(tile-ov)); stride is a float, truncated per tile and
    clamped so that the last tile is flush with the border (utils.py:161-169)."""
    n = math.ceil((extent - overlap) / (tile - overlap))
    step = (extent - tile) / (n - 1) if n > 1 else 0
    return [min(int(i * step), extent - tile) for i in range(n)]


def split_bboxes(w: int, h: int, tile_w: int, tile_h: int, overlap: int) -> List[Tuple[int, int, int, int]]:
    """Row-major (y outer) list of (x, y, tw, th) (utils.py:166-175)."""
    xs = grid_origins(w, tile_w, overlap)
    ys = grid_origins(h, tile_h, overlap)
    return [(x, y, tile_w, tile_h) for y in ys for x in xs]


def init_grid(w: int, h: int, tile_w: int, tile_h: int, overlap: int, tile_bs: int):
    """Clamp + batch exactly like init_grid_bbox (abstractdiffusion.py:176-186).  Note the overlap clamp uses the
    *requested* tile size, not the canvas-clamped one."""
    tw, th = min(tile_w, w), min(tile_h, h)
    ov = max(0, min(overlap, min(tile_w, tile_h) - 4))
    boxes = split_bboxes(w, h, tw, th, ov)
    num_batches = math.ceil(len(boxes) / tile_bs)
    bs = math.ceil(len(boxes) / num_batches)
    batches = [boxes[i * bs:(i + 1) * bs] for i in range(num_batches)]
    return boxes, batches, tw, th, ov


# --------------------------------------------------------------------------------------------------
# maps  (tile_utils/utils.py:180-214)
# --------------------------------------------------------------------------------------------------
def gaussian_weights(tile_w: int, tile_h: int) -> torch.Tensor:
    """utils.py:187-194 -- var 0.01; BOTH axes are normalised by tile_w**2; x midpoint (tw-1)/2, y midpoint th/2;
    float64 outer product, then cast to fp32."""
    var = 0.01

    def prof(t, mid):
        return np.exp(-(t - mid) * (t - mid) / (tile_w * tile_w) / (2 * var)) / np.sqrt(2 * np.pi * var)

    xp = [prof(x, (tile_w - 1) / 2) for x in range(tile_w)]
    yp = [prof(y, tile_h / 2) for y in range(tile_h)]
    return torch.from_numpy(np.outer(yp, xp)).to(torch.float32)


def feather_mask(w: int, h: int, ratio: float) -> torch.Tensor:
    """utils.py:196-214 -- ones; r = int(min(w//2, h//2) * ratio); the four quadrant corners get (d/r)^2 with
    d = min(i, j) < r; odd centre row/column stay 1."""
    m = np.ones((h, w), dtype=np.float32)
    r = int(min(w // 2, h // 2) * ratio)
    for i in range(h // 2):
        for j in range(w // 2):
            d = min(i, j)
            if d >= r:
                continue
            v = (d / r) ** 2
            m[i, j] = m[i, w - 1 - j] = m[h - 1 - i, j] = m[h - 1 - i, w - 1 - j] = v
    return torch.from_numpy(m)


def grid_weight_map(w: int, h: int, boxes: Sequence[Tuple[int, int, int, int]], tile_weight) -> torch.Tensor:
    """utils.py:164-175 -- zeros(1,1,h,w) fp32, `+= init_weight` per tile in list order."""
    m = torch.zeros((1, 1, h, w), dtype=torch.float32)
    for (x, y, tw, th) in boxes:
        m[:, :, y:y + th, x:x + tw] += tile_weight
    return m


# --------------------------------------------------------------------------------------------------
# regions  (abstractdiffusion.py:194-215, utils.py:84-99)
# --------------------------------------------------------------------------------------------------
@dataclass
class Region:
    x: int
    y: int
    w: int
    h: int
    blend_mode: str          # BG | FG
    feather_ratio: float = 0.2

    @property
    def sl(self):
        return (slice(None), slice(None), slice(self.y, self.y + self.h), slice(self.x, self.x + self.w))


def region_rect(W: int, H: int, fx: float, fy: float, fw: float, fh: float) -> Tuple[int, int, int, int]:
    """Fractions -> latent pixels (abstractdiffusion.py:207-214): origin truncated, size ceil'ed, clamped."""
    x, y = max(0, int(fx * W)), max(0, int(fy * H))
    w, h = math.ceil(fw * W), math.ceil(fh * H)
    return x, y, min(W - x, w), min(H - y, h)


# --------------------------------------------------------------------------------------------------
# one model evaluation
# --------------------------------------------------------------------------------------------------
Denoiser = Callable[[torch.Tensor], torch.Tensor]


class BlendOracle:
    """State + arithmetic of AbstractDiffusion/MultiDiffusion/MixtureOfDiffusers restricted to the blend.

    method 'md'  : multidiffusion.py:131-218 (uniform weights, divide where weights > 1 at the end)
    method 'mod' : mixtureofdiffusers.py:29-55, 61-179 (Gaussian weights pre-normalised by 1/weights; no final divide)
    """

    def __init__(self, method: str, W: int, H: int, tile_w: int, tile_h: int, overlap: int, tile_bs: int,
                 regions: Sequence[Region] = (), draw_background: bool = True):
        assert method in ("md", "mod")
        self.method, self.W, self.H = method, W, H
        self.draw_background = draw_background
        self.boxes, self.batches, self.tw, self.th, self.ov = init_grid(W, H, tile_w, tile_h, overlap, tile_bs)
        self.weights = torch.zeros((1, 1, H, W), dtype=torch.float32)          # abstractdiffusion.py:28
        self.tile_weights = gaussian_weights(self.tw, self.th) if method == "mod" else 1.0
        self.weights += grid_weight_map(W, H, self.boxes, self.tile_weights)    # abstractdiffusion.py:181-182
        self.regions = list(regions)
        if self.regions and not draw_background:                                  # abstractdiffusion.py:199-201
            self.weights.zero_()
        self.custom_weights: List[Optional[torch.Tensor]] = []
        self.feather = [feather_mask(r.w, r.h, max(min(r.feather_ratio, 1.0), 0.0)) if r.blend_mode == FG else None
                        for r in self.regions]                                    # utils.py:92-96
        for r in self.regions:
            if r.blend_mode != BG:
                self.custom_weights.append(None)
                continue
            if method == "md":                                                    # multidiffusion.py:44-46
                self.weights[r.sl] += 1.0
                self.custom_weights.append(None)
            else:                                                                 # mixtureofdiffusers.py:50-53
                cw = gaussian_weights(r.w, r.h)
                self.weights[r.sl] += cw
                self.custom_weights.append(cw[None, None])
        if method == "mod":                                                       # mixtureofdiffusers.py:29-36
            self.rescale = 1 / self.weights
            for i, r in enumerate(self.regions):
                if r.blend_mode == BG:
                    self.custom_weights[i] = self.custom_weights[i] * self.rescale[r.sl]

    # -- K2: tile-major batch gather (multidiffusion.py:155, mixtureofdiffusers.py:88,104)
    def gather(self, x_in: torch.Tensor, batch) -> torch.Tensor:
        return torch.cat([x_in[:, :, y:y + th, x:x + tw] for (x, y, tw, th) in batch], dim=0)

    def evaluate(self, x_in: torch.Tensor, tile_fn: Denoiser, region_fn: Optional[Callable] = None, with_boxes: bool = False) -> torch.Tensor:
        """One hijacked forward.  `tile_fn(x_tile[bs*N,C,th,tw])` stands in for the UNet on a tile batch;
        `region_fn(x_region, idx)` for the per-region custom forward.  with_boxes: the callbacks also receive the batch's
        bbox list / the Region (what upstream's repeat_func / custom_func get, multidiffusion.py:163,184) -- used by
        oracle/entry_oracle.py, whose stand-in model depends on per-tile conditioning."""
        N = x_in.shape[0]
        buf = torch.zeros_like(x_in)                                              # abstractdiffusion.py:97-102
        if self.draw_background:
            for batch in self.batches:
                out = tile_fn(self.gather(x_in, batch), batch) if with_boxes else tile_fn(self.gather(x_in, batch))
                for i, (x, y, tw, th) in enumerate(batch):
                    o = out[i * N:(i + 1) * N]
                    if self.method == "md":                                       # multidiffusion.py:166-167
                        buf[:, :, y:y + th, x:x + tw] += o
                    else:                                                         # mixtureofdiffusers.py:122-126
                        wgt = self.tile_weights * self.rescale[:, :, y:y + th, x:x + tw]
                        buf[:, :, y:y + th, x:x + tw] += o * wgt
        fbuf = fmask = fcnt = None
        for i, r in enumerate(self.regions):
            o = region_fn(x_in[r.sl], i, r) if with_boxes else region_fn(x_in[r.sl], i)
            if r.blend_mode == BG:
                if self.method == "md":                                           # multidiffusion.py:189-190
                    buf[r.sl] += o
                else:                                                             # mixtureofdiffusers.py:152-153
                    buf[r.sl] += o * self.custom_weights[i]
            else:                                                                 # :191-198 / :154-161
                if fbuf is None:
                    fbuf = torch.zeros_like(buf)
                    fmask = torch.zeros((1, 1, self.H, self.W))
                    fcnt = torch.zeros((1, 1, self.H, self.W))
                fbuf[r.sl] += o
                fmask[r.sl] += self.feather[i]
                fcnt[r.sl] += 1
        if self.method == "md":                                                   # multidiffusion.py:208
            out = torch.where(self.weights > 1, buf / self.weights, buf)
        else:
            out = buf
        if fbuf is not None:                                                      # :211-216 / :170-175
            fbuf = torch.where(fcnt > 1, fbuf / fcnt, fbuf)
            fmask = torch.where(fcnt > 1, fmask / fcnt, fmask)
            out = torch.where(fcnt > 0, out * (1 - fmask) + fbuf * fmask, out)
        return out


def synthetic_denoiser(x_tile: torch.Tensor) -> torch.Tensor:
    """The stand-in 'UNet' used everywhere in tests/bench: cheap, deterministic, and NOT
    symmetric under a horizontal flip, so tile-order / transposition bugs show up."""
    return 0.9 * x_tile + 0.1 * x_tile.flip(-1)


def synthetic_region_denoiser(x_region: torch.Tensor, idx: int) -> torch.Tensor:
    return (0.8 - 0.05 * idx) * x_region + 0.2 * x_region.flip(-2)


# --------------------------------------------------------------------------------------------------
# per-region initial noise  (scripts/tilediffusion.py:486-529, create_random_tensors_hijack)
# --------------------------------------------------------------------------------------------------
def region_noise(org: torch.Tensor, regions) -> torch.Tensor:
    """org: [N,C,H,W]; regions: [(fx, fy, fw, fh, mode 'Background'|'Foreground', seed)] as fractions of the canvas.
    Restates upstream line by line: per-region CPU-seeded randn, per-kind sum + count, average where count > 1,
    paste background then foreground."""
    H, W = org.shape[2], org.shape[3]
    bg, fg = torch.zeros_like(org), torch.zeros_like(org)
    bgc, fgc = torch.zeros((1, 1, H, W)), torch.zeros((1, 1, H, W))
    for fx, fy, fw, fh, mode, seed in regions:
        x, y = int(fx * W), int(fy * H)
        w, h = math.ceil(fw * W), math.ceil(fh * H)
        x, y = max(0, x), max(0, y)
        w, h = min(W - x, w), min(H - y, h)
        torch.manual_seed(seed)
        r = torch.randn((1, org.shape[1], h, w))
        if mode == "Background":
            bg[:, :, y:y + h, x:x + w] += r
            bgc[:, :, y:y + h, x:x + w] += 1
        else:
            fg[:, :, y:y + h, x:x + w] += r
            fgc[:, :, y:y + h, x:x + w] += 1
    bg = torch.where(bgc > 1, bg / bgc, bg)
    fg = torch.where(fgc > 1, fg / fgc, fg)
    out = torch.where(bgc > 0, bg, org)
    return torch.where(fgc > 0, fg, out)


# --------------------------------------------------------------------------------------------------
# Noise Inversion: renoise composite of sample_img2img  (tile_methods/abstractdiffusion.py:651-676)
# --------------------------------------------------------------------------------------------------
def noise_inverse_blend(noise: torch.Tensor, inverse_noise: torch.Tensor, renoise_mask: Optional[torch.Tensor], regions=(),
                        enable_grid_bbox: bool = True) -> torch.Tensor:
    """noise / inverse_noise [N,C,H,W]; renoise_mask [H,W] (already strength-scaled and clamped, :612-616) or None;
    regions: [Region] (rect in latent px, mode, feather ratio).  Restates :655-676 op by op:
    without a mask the inverse noise is used as is; with the grid disabled the job's noise is first re-weighted by the
    background hit count and the count-averaged foreground feather masks."""
    if renoise_mask is None:                                                      # :675-676
        return inverse_noise
    if not enable_grid_bbox:                                                      # :658-672
        H, W = noise.shape[2], noise.shape[3]
        background_count = torch.zeros((1, 1, H, W))
        foreground_noise = torch.zeros_like(noise)
        foreground_weight = torch.zeros((1, 1, H, W))
        foreground_count = torch.zeros((1, 1, H, W))
        for r in regions:
            if r.blend_mode == BG:
                background_count[r.sl] += 1
            elif r.blend_mode == FG:
                foreground_noise[r.sl] += noise[r.sl]
                foreground_weight[r.sl] += feather_mask(r.w, r.h, r.feather_ratio)
                foreground_count[r.sl] += 1
        background_noise = torch.where(background_count > 0, noise, 0)
        foreground_noise = torch.where(foreground_count > 0, foreground_noise / foreground_count, 0)
        foreground_weight = torch.where(foreground_count > 0, foreground_weight / foreground_count, 0)
        noise = background_noise * (1 - foreground_weight) + foreground_noise * foreground_weight
    return ((1 - renoise_mask) * inverse_noise + renoise_mask * noise) / ((renoise_mask ** 2 + (1 - renoise_mask) ** 2) ** 0.5)
