"""
ORACLE (test infrastructure only) -- CPU restatement of upstream's DemoFusion model evaluation
(tile_methods/demofusion.py:93-162 window / view construction, :164-178 Gaussian filter, :219-324 sample_one_step).

Only tests/ may import this file.  Parity status: PINNED by tests/test_oracle_vs_reference.py::test_demofusion_* (runs the upstream
DemoFusion delegate itself under hostsim/stub_host.py with the same `random` seed and compares bit for bit).
"""
from __future__ import annotations

import math
import random
from typing import Callable, List, Tuple

import torch
import torch.nn.functional as F


def jitter_range(w: int, h: int, window: int, overlap: int, stride: int, random_jitter: bool) -> int:
    """demofusion.py:121 -- min(max((min(w, h) - stride) // 4, 0), min(window // 2, overlap // 2)); 0 without jitter."""
    if not random_jitter:
        return 0
    return min(max((min(w, h) - stride) // 4, 0), min(int(window / 2), int(overlap / 2)))


def views(w: int, h: int, window: int, overlap: int, random_jitter: bool) -> Tuple[List[Tuple[int, int]], int, int, int]:
    """get_views + split_bboxes_jitter (:93-148): window origins (x, y) in the JITTER-PADDED canvas, row-major; uses the module
    `random` exactly like upstream (seed it before the call).  Returns (origins, jitter range J, overlap, stride)."""
    overlap = max(0, min(overlap, window - 4))
    stride = max(4, window - overlap)
    cols = math.ceil((w - overlap) / (window - overlap)) or 1
    rows = math.ceil((h - overlap) / (window - overlap)) or 1
    dx = (w - window) / (cols - 1) if cols > 1 else 0
    dy = (h - window) / (rows - 1) if rows > 1 else 0
    J = 0
    out = []
    for row in range(rows):
        for col in range(cols):
            y = min(int(row * dy), h - window)
            x = min(int(col * dx), w - window)
            if random_jitter:
                J = jitter_range(w, h, window, overlap, stride, True)
                xj = yj = 0
                if x != 0 and x + window != w:
                    xj = random.randint(-J, J)
                elif x == 0 and x + window != w:
                    xj = random.randint(-J, 0)
                elif x != 0 and x + window == w:
                    xj = random.randint(0, J)
                if y != 0 and y + window != h:
                    yj = random.randint(-J, J)
                elif y == 0 and y + window != h:
                    yj = random.randint(-J, 0)
                elif y != 0 and y + window == h:
                    yj = random.randint(0, J)
                y += yj + J
                x += xj + J
            out.append((x, y))
    return out, J, overlap, stride


def batches(items: list, bs: int) -> List[list]:
    nb = math.ceil(len(items) / bs)
    k = math.ceil(len(items) / nb)
    return [items[i * k:(i + 1) * k] for i in range(nb)]


def gaussian_kernel(kernel_size: int, sigma, channels: int) -> torch.Tensor:
    """:164-171"""
    xc = torch.arange(kernel_size)
    g1 = torch.exp(-(xc - (kernel_size - 1) / 2) ** 2 / (2 * sigma ** 2))
    g1 = g1 / g1.sum()
    g2 = g1[:, None] * g1[None, :]
    return g2[None, None, :, :].repeat(channels, 1, 1, 1)


def gaussian_filter(latents: torch.Tensor, kernel_size: int, sigma) -> torch.Tensor:
    """:173-178"""
    c = latents.shape[1]
    k = gaussian_kernel(kernel_size, sigma, c).to(latents.dtype)
    return F.conv2d(latents, k, padding=kernel_size // 2, groups=c)


def sample_one_step(x_in: torch.Tensor, origins, window: int, J: int, tile_bs: int, global_bs: int, S: int, mixture: bool,
                    use_gaussian: bool, sig: float, cosine_factor: torch.Tensor, cosine_scale_2: float, cosine_scale_3: float,
                    tile_fn: Callable) -> torch.Tensor:
    """:219-324.  x_in: [N, C, H + 2J, W + 2J] (already jitter-padded).  tile_fn(x_tile) stands for the model call on a tile batch.
    S = p.current_scale_num.  Quirk kept: the dilated views end at  x.shape[3] - J  on BOTH axes (:262)."""
    N = x_in.shape[0]
    buf = torch.zeros_like(x_in)
    wts = torch.zeros_like(x_in)
    for batch in batches(list(origins), tile_bs):
        x_tile = torch.cat([x_in[:, :, y:y + window, x:x + window] for (x, y) in batch], dim=0)
        out = tile_fn(x_tile)
        for i, (x, y) in enumerate(batch):
            buf[:, :, y:y + window, x:x + window] += out[i * N:(i + 1) * N]
            wts[:, :, y:y + window, x:x + window] += 1
    wts = torch.where(wts == 0, torch.tensor(1), wts)
    x_local = buf / wts
    buf = torch.zeros_like(buf)
    wts = torch.zeros_like(wts)

    std_, mean_ = x_in.std(), x_in.mean()
    c3 = 0.99 * cosine_factor ** cosine_scale_3 + 1e-2
    x_in_g = None
    if use_gaussian:
        x_in_g = gaussian_filter(x_in, kernel_size=(2 * S - 1), sigma=sig * c3)
        x_in_g = (x_in_g - x_in_g.mean()) / x_in_g.std() * std_ + mean_
    x_global = torch.zeros_like(x_local)
    end = x_global.shape[3] - J
    cells = [(x, y) for y in range(S) for x in range(S)]
    cells = cells + cells if mixture else cells
    total = len(cells)
    gb = batches(cells, global_bs)
    gbs = len(gb[0])
    cur = 0

    def dil(t, bx, by):
        return t[:, :, by + J:end:S, bx + J:end:S]

    for batch in gb:
        cur += len(batch)
        if mixture:
            if cur > total // 2 and cur - gbs < total // 2:
                res = len(batch) - (cur - total // 2)
                xi = torch.cat([dil(x_in, bx, by) if idx < res else dil(x_in_g, bx, by) for idx, (bx, by) in enumerate(batch)], dim=0)
            elif cur > total // 2:
                xi = torch.cat([dil(x_in_g, bx, by) for (bx, by) in batch], dim=0)
            else:
                xi = torch.cat([dil(x_in, bx, by) for (bx, by) in batch], dim=0)
        else:
            xi = torch.cat([dil(x_in_g, bx, by) for (bx, by) in batch], dim=0)
        out = tile_fn(xi)
        for idx, (bx, by) in enumerate(batch):
            x_global[:, :, by + J:end:S, bx + J:end:S] += out[idx * N:(idx + 1) * N]
    if mixture:
        buf += x_global / 2
    else:
        buf += x_global
    wts += 1
    x_global = buf / wts
    c2 = cosine_factor ** cosine_scale_2
    return x_local * (1 - c2) + x_global * c2
