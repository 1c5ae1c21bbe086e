"""
Mixture-of-Diffusers delegate (https://github.com/albarji/mixture-of-diffusers) on the mdtile engine.

Same hijack surface as upstream tile_methods/mixtureofdiffusers.py: `shared.sd_model.apply_model` is replaced by
`apply_model_hijack`, which blends per-tile eps predictions with Gaussian tile weights that were pre-normalised by the
weight-sum map (`rescale_factor = 1 / weights`, upstream :29-36).  The per-tile `w = tile_weights * rescale[slicer]`,
`x_buffer[slicer] += out * w` loop (:122-126), Gaussian background regions (:152-153) and the feather composite
(:154-175) run as ONE mdtile_blend launch; there is no final division (:177-179).
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor

from modules import devices, extra_networks, shared
from modules.shared import state

import mdtile
from tile_methods.abstractdiffusion import AbstractDiffusion
from tile_utils.utils import BlendMode, Condition, gaussian_weights


class MixtureOfDiffusers(AbstractDiffusion):

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.custom_weights: List[Optional[Tensor]] = []   # per region: Gaussian [1,1,h,w] (background) or None
        self.get_weight = gaussian_weights
        self.rescale_factor: Optional[Tensor] = None

    # ---- hijack ---------------------------------------------------------------------------------------------------
    def hook(self):
        if not hasattr(shared.sd_model, "apply_model_original_md"):
            shared.sd_model.apply_model_original_md = shared.sd_model.apply_model
        shared.sd_model.apply_model = self.apply_model_hijack

    @staticmethod
    def unhook():
        if hasattr(shared.sd_model, "apply_model_original_md"):
            shared.sd_model.apply_model = shared.sd_model.apply_model_original_md
            del shared.sd_model.apply_model_original_md

    # ---- weights --------------------------------------------------------------------------------------------------
    def get_tile_weights(self) -> Tensor:
        if not hasattr(self, "tile_weights"):
            self.tile_weights = self.get_weight(self.tile_w, self.tile_h)
        return self.tile_weights

    def init_custom_bbox(self, *args):
        super().init_custom_bbox(*args)
        for bbox in self.custom_bboxes:
            if bbox.blend_mode == BlendMode.BACKGROUND:
                cw = self.get_weight(bbox.w, bbox.h)
                mdtile.weight_map_add_rect(self.weights, bbox.x, bbox.y, bbox.w, bbox.h, cw)
                self.custom_weights.append(cw.unsqueeze(0).unsqueeze(0))
            else:
                self.custom_weights.append(None)

    def init_done(self):
        super().init_done()
        # Gaussian weights can be ~1e-10: normalise up front (inf where nothing is painted, exactly as upstream)
        self.rescale_factor = mdtile.reciprocal(self.weights)
        for bbox_id, bbox in enumerate(self.custom_bboxes):
            if bbox.blend_mode == BlendMode.BACKGROUND:
                mdtile.rect_mul_canvas(self.custom_weights[bbox_id], self.rescale_factor, bbox.x, bbox.y, bbox.w, bbox.h)

    def region_weights(self):
        return [b.feather_mask if b.blend_mode == BlendMode.FOREGROUND else self.custom_weights[i]
                for i, b in enumerate(self.custom_bboxes)]

    # ---- the hot path ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def apply_model_hijack(self, x_in: Tensor, t_in: Tensor, cond, noise_inverse_step: int = -1):
        c_in = cond
        N, C, H, W = x_in.shape
        if (H, W) != (self.h, self.w):
            self.reset_controlnet_tensors()
            return shared.sd_model.apply_model_original_md(x_in, t_in, c_in)
        x_in = x_in.contiguous()

        tile_outs: List[Tensor] = []
        if self.draw_background:
            x_tiles = mdtile.gather_all(self.plan, x_in)                 # K2, one launch
            for batch_id, bboxes in enumerate(self.batched_bboxes):
                if state.interrupted:
                    return x_in
                n = len(bboxes)
                t_tile = torch.cat([t_in] * n, dim=0)
                if isinstance(c_in, dict):
                    tcond = torch.cat([self.get_tcond(c_in)] * n, dim=0)
                    icond = torch.cat([self.slice_icond(self.get_icond(c_in), b) for b in bboxes], dim=0)
                    vc = self.get_vcond(c_in)
                    vcond = torch.cat([vc] * n, dim=0) if vc is not None else None
                    c_tile = self.make_cond_dict(c_in, tcond, icond, vcond)
                else:
                    print(">> [WARN] not supported, make an issue on github!!")
                    c_tile = c_in
                self.switch_controlnet_tensors(batch_id, N, n, is_denoise=True)
                self.switch_stablesr_tensors(batch_id)
                out = shared.sd_model.apply_model_original_md(x_tiles[batch_id], t_tile, c_tile)
                tile_outs.append(out.to(x_in.dtype).contiguous())
                self.update_pbar()

        region_outs: List[Tensor] = []
        for bbox_id, bbox in enumerate(self.custom_bboxes):
            if not self.p.disable_extra_networks:
                with devices.autocast():
                    extra_networks.activate(self.p, bbox.extra_network_data)
            x_tile = mdtile.gather_rect(x_in, bbox.x, bbox.y, bbox.w, bbox.h)
            if noise_inverse_step < 0:
                out = self.custom_apply_model(x_tile, t_in, c_in, bbox_id, bbox)
            else:
                tcond = Condition.reconstruct_cond(bbox.cond, noise_inverse_step)
                icond = self.slice_icond(self.get_icond(c_in), bbox)
                c_out = self.make_cond_dict(c_in, tcond, icond, self.get_vcond(c_in))
                out = shared.sd_model.apply_model(x_tile, t_in, cond=c_out)
            region_outs.append(out.to(x_in.dtype).contiguous())
            self.update_pbar()
            if not self.p.disable_extra_networks:
                with devices.autocast():
                    extra_networks.deactivate(self.p, bbox.extra_network_data)

        # K4 + K6 + K7 in one launch; pixels nobody paints stay 0, as upstream
        self.x_buffer = mdtile.blend(self.blend_plan(), mdtile.METHOD_MOD, tile_outs, N, C, tile_w=self.get_tile_weights() if tile_outs else None,
                                     rescale=self.rescale_factor, regions=self.region_specs(region_outs),
                                     dtype=x_in.dtype, device=x_in.device)
        return self.x_buffer

    def custom_apply_model(self, x_in, t_in, c_in, bbox_id, bbox) -> Tensor:
        if self.is_kdiff:
            return self.kdiff_custom_forward(x_in, t_in, c_in, bbox_id, bbox,
                                             forward_func=shared.sd_model.apply_model_original_md)

        def forward_func(x, c, ts, unconditional_conditioning, *args, **kwargs) -> Tensor:
            merged = {}
            for key in c:   # cond + uncond batched the way p_sample_ddim does
                if isinstance(c[key], list):
                    merged[key] = [torch.cat([unconditional_conditioning[key][i], c[key][i]]) for i in range(len(c[key]))]
                else:
                    merged[key] = torch.cat([unconditional_conditioning[key], c[key]])
            return shared.sd_model.apply_model_original_md(x, ts, merged)

        return self.ddim_custom_forward(x_in, c_in, bbox, ts=t_in, forward_func=forward_func)

    @torch.no_grad()
    def get_noise(self, x_in: Tensor, sigma_in: Tensor, cond_in, step: int) -> Tensor:
        return self.apply_model_hijack(x_in, sigma_in, cond=cond_in, noise_inverse_step=step)
