"""
DemoFusion delegate (https://arxiv.org/abs/2311.16973) on the mdtile engine -- same hijack surface as upstream
tile_methods/demofusion.py (hook / forward_one_step / sample_one_step / get_views / get_noise, the attributes the Script pokes:
window_size, sig, jitter_range, batched_bboxes, global_batched_bboxes ...).

One hijacked model evaluation (upstream :219-324) is
    local path   jittered, equally sized windows  -> mdtile_gather_rects per batch, the model, ONE mdtile_window_blend (count average)
    global path  S x S dilated views of the Gaussian-filtered latent -> mdtile_depthwise_blur + mdtile_restandardize,
                 mdtile_dilated_gather per batch, the model
    mix          mdtile_demofusion_combine: scatter of the global outputs onto the lattice + x_local * (1 - c2) + x_global * c2
instead of upstream's per-window slice / `+=` / count / where / div chains and per-cell strided `+=`.
"""
from __future__ import annotations

import math
import random
from typing import List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from modules import devices, shared
from modules.shared import state

import mdtile
from tile_methods.abstractdiffusion import AbstractDiffusion
from tile_utils.utils import BBox


class DemoFusion(AbstractDiffusion):

    def __init__(self, p, *args, **kwargs):
        super().__init__(p, *args, **kwargs)
        assert p.sampler_name != "UniPC", "Demofusion is not compatible with UniPC!"
        self.jitter_range = 0
        self.window_size = None
        self.sig = 0.0
        self.repeat_3 = False
        self.windows: mdtile.WindowSet = None

    # ---- hijack (upstream :21-42) ------------------------------------------------------------------------------------------
    def hook(self):
        from modules import sd_samplers_common
        steps, self.t_enc = sd_samplers_common.setup_img2img_steps(self.p, None)
        cfg = self.sampler.model_wrap_cfg
        cfg.forward_ori = cfg.forward
        self.sampler_forward = cfg.inner_model.forward
        cfg.forward = self.forward_one_step
        if not self.is_kdiff:
            self.timesteps = self.sampler.get_timesteps(self.p, steps)

    @staticmethod
    def unhook():
        if hasattr(shared.sd_model, "apply_model_ori"):
            shared.sd_model.apply_model = shared.sd_model.apply_model_ori
            del shared.sd_model.apply_model_ori

    # ---- cond batching -------------------------------------------------------------------------------------------------------
    def repeat_tensor(self, x: Tensor, n: int) -> Tensor:
        if n == 1:
            return x
        tail = x.dim() - 1
        if x.shape[0] == 1:
            return x.expand([n] + [-1] * tail)
        return x.repeat([n] + [1] * tail)

    def repeat_cond_dict(self, cond_in, bboxes, mode: int):
        """mode 0: local windows (BBox list); mode 1: lattice cells ((bx, by) list) -- upstream :59-86."""
        n = len(bboxes)
        tcond = self.repeat_tensor(self.get_tcond(cond_in), n)
        icond = self.get_icond(cond_in)
        if tuple(icond.shape[2:]) == (self.h, self.w):      # img2img: the image conditioning follows the views
            if mode == 0:
                if self.p.random_jitter:
                    j = self.jitter_range
                    icond = F.pad(icond, (j, j, j, j), "constant", value=0)
                icond = mdtile.gather_rects(icond.contiguous(), [(b.x, b.y) for b in bboxes], bboxes[0].w, bboxes[0].h)
            else:
                S = self.p.current_scale_num
                icond = torch.cat([icond[:, :, by::S, bx::S] for (bx, by) in bboxes], dim=0)
        else:
            icond = self.repeat_tensor(icond, n)
        vcond = self.get_vcond(cond_in)
        if vcond is not None:
            vcond = self.repeat_tensor(vcond, n)
        return self.make_cond_dict(cond_in, tcond, icond, vcond)

    # ---- views (upstream :89-162) -----------------------------------------------------------------------------------------
    def global_split_bboxes(self) -> List[Tuple[int, int]]:
        S = self.p.current_scale_num
        cells = [(x, y) for y in range(S) for x in range(S)]
        return cells + cells if self.p.mixture else cells

    def split_bboxes_jitter(self, w_l: int, h_l: int, tile_w: int, tile_h: int, overlap: int = 16, init_weight=1.0):
        """Window origins on a regular grid, each moved by a random offset of at most the jitter range and shifted into the
        jitter-padded canvas; draws from the module `random` in upstream's order (so a seeded run lands on the same windows)."""
        cols = math.ceil((w_l - overlap) / (tile_w - overlap)) or 1
        rows = math.ceil((h_l - overlap) / (tile_h - overlap)) or 1
        dx = (w_l - tile_w) / (cols - 1) if cols > 1 else 0
        dy = (h_l - tile_h) / (rows - 1) if rows > 1 else 0
        self.jitter_range = 0
        self._nomx = [min(int(c * dx), w_l - tile_w) for c in range(cols)]
        self._nomy = [min(int(r * dy), h_l - tile_h) for r in range(rows)]
        out: List[BBox] = []
        for r in range(rows):
            for c in range(cols):
                y, x = self._nomy[r], self._nomx[c]
                if self.p.random_jitter:
                    self.jitter_range = min(max((min(self.w, self.h) - self.stride) // 4, 0), min(int(self.window_size / 2), int(self.overlap / 2)))
                    J = self.jitter_range
                    xj = yj = 0
                    if x != 0 and x + tile_w != w_l:
                        xj = random.randint(-J, J)
                    elif x == 0 and x + tile_w != w_l:
                        xj = random.randint(-J, 0)
                    elif x != 0 and x + tile_w == w_l:
                        xj = random.randint(0, J)
                    if y != 0 and y + tile_h != h_l:
                        yj = random.randint(-J, J)
                    elif y == 0 and y + tile_h != h_l:
                        yj = random.randint(-J, 0)
                    elif y != 0 and y + tile_h == h_l:
                        yj = random.randint(0, J)
                    y += yj + J
                    x += xj + J
                out.append(BBox(x, y, tile_w, tile_h))
        return out, None

    def get_views(self, overlap: int, tile_bs: int, tile_bs_g: int):
        self.enable_grid_bbox = True
        self.tile_w = self.tile_h = self.window_size
        self.overlap = max(0, min(overlap, self.window_size - 4))
        self.stride = max(4, self.window_size - self.overlap)
        bboxes, _ = self.split_bboxes_jitter(self.w, self.h, self.tile_w, self.tile_h, self.overlap, self.get_tile_weights())
        self.num_tiles = len(bboxes)
        self.num_batches = math.ceil(self.num_tiles / tile_bs)
        self.tile_bs = math.ceil(len(bboxes) / self.num_batches)
        self.batched_bboxes = [bboxes[i * self.tile_bs:(i + 1) * self.tile_bs] for i in range(self.num_batches)]
        self.windows = mdtile.WindowSet([(b.x, b.y) for b in bboxes], self._nomx, self._nomy, self.jitter_range, self.window_size, devices.device)
        cells = self.global_split_bboxes()
        self.global_num_tiles = len(cells)
        self.global_num_batches = math.ceil(self.global_num_tiles / tile_bs_g)
        self.global_tile_bs = math.ceil(len(cells) / self.global_num_batches)
        self.global_batched_bboxes = [cells[i * self.global_tile_bs:(i + 1) * self.global_tile_bs] for i in range(self.global_num_batches)]

    # ---- Gaussian filter (upstream :164-178) --------------------------------------------------------------------------------
    def gaussian_kernel(self, kernel_size=3, sigma=1.0, channels=3) -> Tensor:
        xc = torch.arange(kernel_size, device=devices.device)
        g1 = torch.exp(-(xc - (kernel_size - 1) / 2) ** 2 / (2 * sigma ** 2))
        g1 = g1 / g1.sum()
        g2 = g1[:, None] * g1[None, :]
        return g2[None, None, :, :].repeat(channels, 1, 1, 1)

    def gaussian_filter(self, latents: Tensor, kernel_size=3, sigma=1.0) -> Tensor:
        k = self.gaussian_kernel(kernel_size, sigma, 1)[0, 0].to(device=latents.device, dtype=torch.float32).contiguous()
        return mdtile.depthwise_blur(latents.contiguous(), k)

    # ---- one sampler step (upstream :183-216) ----------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_one_step(self, x_in, sigma, **kwarg):
        p = self.p
        if self.is_kdiff:
            x_noisy = p.x + p.noise * sigma[0]
        else:
            ac = p.sd_model.alphas_cumprod
            t = self.timesteps[self.t_enc - p.current_step]
            x_noisy = p.x * torch.sqrt(ac[t]) + p.noise * torch.sqrt(1 - ac[t])
        self.cosine_factor = 0.5 * (1 + torch.cos(torch.pi * torch.tensor((p.current_step + 1) / (self.t_enc + 1))))
        c1 = self.cosine_factor ** p.cosine_scale_1
        x_in = x_in * (1 - c1) + x_noisy * c1
        j = self.jitter_range if p.random_jitter else 0
        x_pad = F.pad(x_in, (j, j, j, j), "constant", value=0)
        _, _, H, W = x_in.shape
        cfg = self.sampler.model_wrap_cfg
        cfg.inner_model.forward = self.sample_one_step
        self.repeat_3 = False
        x_out = cfg.forward_ori(x_pad, sigma, **kwarg)
        cfg.inner_model.forward = self.sampler_forward
        return x_out[:, :, j:j + H, j:j + W]

    # ---- one model evaluation (upstream :219-324) --------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_one_step(self, x_in, sigma, cond):
        p = self.p

        def repeat_func_1(x_tile, bboxes, mode=0):
            return self.sampler_forward(x_tile, self.repeat_tensor(sigma, len(bboxes)), cond=self.repeat_cond_dict(cond, bboxes, mode))

        def repeat_func_2(x_tile, bboxes, mode=0):
            n = len(bboxes)
            cond_tile = self.repeat_cond_dict(cond, bboxes, mode) if isinstance(cond, dict) else self.repeat_tensor(cond, n)
            return self.sampler_forward(x_tile, self.repeat_tensor(sigma, n), cond=cond_tile)

        def repeat_func_3(x_tile, bboxes, mode=0):
            return shared.sd_model.apply_model(x_tile, sigma.repeat(len(bboxes)), cond=self.repeat_cond_dict(cond, bboxes, mode))

        if self.repeat_3:
            repeat_func, self.repeat_3 = repeat_func_3, False
        else:
            repeat_func = repeat_func_1 if self.is_kdiff else repeat_func_2
        x_in = x_in.contiguous()
        N, C, Hp, Wp = x_in.shape
        win, J, S = self.window_size, self.jitter_range, p.current_scale_num

        # local path: every window batch through the model, then ONE count-averaged blend
        outs = []
        for bboxes in self.batched_bboxes:
            if state.interrupted:
                return x_in
            x_tile = mdtile.gather_rects(x_in, [(b.x, b.y) for b in bboxes], win, win)
            outs.append(repeat_func(x_tile, bboxes).to(x_in.dtype))
        x_local = mdtile.window_blend(torch.cat(outs, dim=0).contiguous(), self.windows, N, C, Hp, Wp)

        # Gaussian-filtered latent, re-standardised to the statistics of x_in (:259-264)
        x_in_g = None
        if p.gaussian_filter:
            tgt = mdtile.moments(x_in)
            c3 = 0.99 * self.cosine_factor ** p.cosine_scale_3 + 1e-2
            x_in_g = self.gaussian_filter(x_in, kernel_size=(2 * S - 1), sigma=self.sig * c3)
            cur = mdtile.moments(x_in_g)
            x_in_g = mdtile.restandardize(x_in_g, torch.cat([cur, tgt]).float())

        # global path: lattice cells in list order (mixture: first the cells of x_in, then the same cells of the filtered latent)
        if not hasattr(p.sd_model, "apply_model_ori"):
            p.sd_model.apply_model_ori = p.sd_model.apply_model
        p.sd_model.apply_model = self.apply_model_hijack
        end = Wp - J                                   # upstream takes the end of BOTH axes from the width (:262)
        row_end = min(end, Hp)
        h0s = {math.ceil((row_end - by - J) / S) for by in range(S)}
        w0s = {math.ceil((end - bx - J) / S) for bx in range(S)}
        if len(h0s) != 1 or len(w0s) != 1:
            raise ValueError(f"DemoFusion: the {S}x{S} lattice does not tile a {Hp - 2 * J}x{Wp - 2 * J} latent evenly")
        h0, w0 = h0s.pop(), w0s.pop()
        total, cur_n, gouts = self.global_num_tiles, 0, []
        for cells in self.global_batched_bboxes:
            cur_n += len(cells)
            if p.mixture:
                if cur_n > total // 2 and cur_n - self.global_tile_bs < total // 2:
                    n_from_x = len(cells) - (cur_n - total // 2)
                elif cur_n > total // 2:
                    n_from_x = 0
                else:
                    n_from_x = len(cells)
            else:
                n_from_x = 0
            src_x = x_in if n_from_x > 0 else x_in_g
            x_g_tiles = mdtile.dilated_gather(src_x, x_in_g, n_from_x, cells, S, J, h0, w0)
            gouts.append(repeat_func(x_g_tiles, cells, mode=1).to(x_in.dtype))
        p.sd_model.apply_model = p.sd_model.apply_model_ori

        c2 = float(self.cosine_factor ** p.cosine_scale_2)
        self.x_buffer = mdtile.demofusion_combine(x_local, torch.cat(gouts, dim=0).contiguous(), S, J, bool(p.mixture), c2)
        return self.x_buffer

    @torch.no_grad()
    def apply_model_hijack(self, x_in: Tensor, t_in: Tensor, cond):
        return self.p.sd_model.apply_model_ori(x_in, t_in, cond)

    def get_noise(self, x_in: Tensor, sigma_in: Tensor, cond_in, step: int) -> Tensor:
        """Noise-inversion entry point (upstream :341-349)."""
        cond_org = cond_in.copy()
        self.repeat_3 = True
        self.cosine_factor = 0.5 * (1 + torch.cos(torch.pi * torch.tensor((self.p.current_step + 1) / (self.t_enc + 1))))
        j = self.jitter_range
        _, _, H, W = x_in.shape
        x_pad = F.pad(x_in, (j, j, j, j), "constant", value=0)
        return self.sample_one_step(x_pad, sigma_in, cond_org)[:, :, j:j + H, j:j + W]
