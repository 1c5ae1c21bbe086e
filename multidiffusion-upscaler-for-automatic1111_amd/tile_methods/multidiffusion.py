"""
MultiDiffusion delegate (https://arxiv.org/abs/2302.08113) on the mdtile engine.

Same hijack surface as upstream tile_methods/multidiffusion.py (hook / kdiff_forward / ddim_forward / sample_one_step /
repeat_tensor / repeat_cond_dict / get_noise), but one model evaluation now costs
    1 gather launch  +  the UNet calls  +  1 blend launch
instead of upstream's ~3T+5 eager kernels (per-tile slice+cat, per-tile strided `+=`, zero/gt/div/where):
the overlap accumulation (:166-167), background regions (:189-190), the `where(weights > 1, buf / weights, buf)`
normalisation (:208) and the foreground feather composite (:191-198, :211-216) are all inside mdtile_blend.
"""
from __future__ import annotations

from typing import Callable, List

import torch
from torch import Tensor

from modules import devices, extra_networks, shared
from modules.shared import state

import mdtile
from tile_methods.abstractdiffusion import AbstractDiffusion
from tile_utils.utils import BlendMode, Condition, CustomBBox


class MultiDiffusion(AbstractDiffusion):

    def __init__(self, p, *args, **kwargs):
        super().__init__(p, *args, **kwargs)
        assert p.sampler_name != "UniPC", "MultiDiffusion is not compatible with UniPC!"

    # ---- hijack ---------------------------------------------------------------------------------------------------
    def hook(self):
        inner = self.sampler.model_wrap_cfg.inner_model
        self.sampler_forward = inner.forward
        inner.forward = self.kdiff_forward if self.is_kdiff else self.ddim_forward

    @staticmethod
    def unhook():
        pass  # the hijacked sampler dies with the job

    def init_custom_bbox(self, *args):
        super().init_custom_bbox(*args)
        for bbox in self.custom_bboxes:  # a background region counts as one more uniform-weight layer
            if bbox.blend_mode == BlendMode.BACKGROUND:
                mdtile.weight_map_add_rect(self.weights, bbox.x, bbox.y, bbox.w, bbox.h, None, 1.0)

    @torch.no_grad()
    def kdiff_forward(self, x_in: Tensor, sigma_in: Tensor, cond) -> Tensor:
        def org_func(x):
            return self.sampler_forward(x, sigma_in, cond=cond)

        def repeat_func(x_tile, bboxes):
            return self.sampler_forward(x_tile, self.repeat_tensor(sigma_in, len(bboxes)),
                                        cond=self.repeat_cond_dict(cond, bboxes))

        def custom_func(x, bbox_id, bbox):
            return self.kdiff_custom_forward(x, sigma_in, cond, bbox_id, bbox, self.sampler_forward)

        return self.sample_one_step(x_in, org_func, repeat_func, custom_func)

    @torch.no_grad()
    def ddim_forward(self, x_in: Tensor, ts_in: Tensor, cond) -> Tensor:
        def org_func(x):
            return self.sampler_forward(x, ts_in, cond=cond)

        def repeat_func(x_tile, bboxes):
            n = len(bboxes)
            cond_tile = self.repeat_cond_dict(cond, bboxes) if isinstance(cond, dict) else self.repeat_tensor(cond, n)
            return self.sampler_forward(x_tile, self.repeat_tensor(ts_in, n), cond=cond_tile)

        def custom_func(x, bbox_id, bbox):
            def forward_func(x, *args, **kwargs):          # the region's control tensors are set right before its model call (:92-95)
                self.set_custom_controlnet_tensors(bbox_id, 2 * x.shape[0])
                self.set_custom_stablesr_tensors(bbox_id)
                return self.sampler_forward(x, *args, **kwargs)
            return self.ddim_custom_forward(x, cond, bbox, ts_in, forward_func)

        return self.sample_one_step(x_in, org_func, repeat_func, custom_func)

    # ---- cond batching (host-side, no arithmetic) --------------------------------------------------------------------
    def repeat_tensor(self, x: Tensor, n: int) -> Tensor:
        """Repeat along dim 0 for a batch of n tiles (a view when the source batch is 1)."""
        if n == 1:
            return x
        tail = x.dim() - 1
        if x.shape[0] == 1:
            return x.expand([n] + [-1] * tail)
        return x.repeat([n] + [1] * tail)

    def repeat_cond_dict(self, cond_in, bboxes: List[CustomBBox]):
        n = len(bboxes)
        tcond = self.repeat_tensor(self.get_tcond(cond_in), n)
        icond = self.get_icond(cond_in)
        if tuple(icond.shape[2:]) == (self.h, self.w):   # img2img: the image conditioning is tiled like the latent
            icond = torch.cat([icond[b.slicer] for b in bboxes], dim=0)
        else:
            icond = self.repeat_tensor(icond, n)
        vcond = self.get_vcond(cond_in)
        if vcond is not None:
            vcond = self.repeat_tensor(vcond, n)
        return self.make_cond_dict(cond_in, tcond, icond, vcond)

    # ---- the hot path ----------------------------------------------------------------------------------------------
    def sample_one_step(self, x_in: Tensor, org_func: Callable, repeat_func: Callable, custom_func: Callable) -> Tensor:
        """One hijacked model evaluation over the whole latent.
            x_in        [N, C, H, W] current latent (N = cond + uncond copies)
            org_func    untiled forward (used when the size does not match, e.g. the hires pass)
            repeat_func forward for one tile batch  [bs*N, C, th, tw] -> same shape
            custom_func forward for one custom region
        """
        N, C, H, W = x_in.shape
        if (H, W) != (self.h, self.w):
            self.reset_controlnet_tensors()
            return org_func(x_in)
        x_in = x_in.contiguous()

        tile_outs: List[Tensor] = []
        if self.draw_background:
            x_tiles = mdtile.gather_all(self.plan, x_in)                 # K2: every tile batch, one launch
            for batch_id, bboxes in enumerate(self.batched_bboxes):
                if state.interrupted:
                    return x_in
                self.switch_controlnet_tensors(batch_id, N, len(bboxes))
                self.switch_stablesr_tensors(batch_id)
                out = repeat_func(x_tiles[batch_id], bboxes)
                tile_outs.append(out.to(x_in.dtype).contiguous())
                self.update_pbar()

        region_outs: List[Tensor] = []
        for bbox_id, bbox in enumerate(self.custom_bboxes):
            if state.interrupted:
                return x_in
            if not self.p.disable_extra_networks:
                with devices.autocast():
                    extra_networks.activate(self.p, bbox.extra_network_data)
            x_tile = mdtile.gather_rect(x_in, bbox.x, bbox.y, bbox.w, bbox.h)
            region_outs.append(custom_func(x_tile, bbox_id, bbox).to(x_in.dtype).contiguous())
            if not self.p.disable_extra_networks:
                with devices.autocast():
                    extra_networks.deactivate(self.p, bbox.extra_network_data)
            self.update_pbar()

        # K3 + K5 + K6 + K7 in one launch
        self.x_buffer = mdtile.blend(self.blend_plan(), mdtile.METHOD_MD, tile_outs, N, C, weights=self.weights,
                                     regions=self.region_specs(region_outs), dtype=x_in.dtype, device=x_in.device)
        return self.x_buffer

    def get_noise(self, x_in: Tensor, sigma_in: Tensor, cond_in, step: int) -> Tensor:
        """Noise-inversion entry point: the same blend over `apply_model` outputs."""
        cond_org = cond_in.copy()

        def org_func(x):
            return shared.sd_model.apply_model(x, sigma_in, cond=cond_org)

        def repeat_func(x_tile, bboxes):
            return shared.sd_model.apply_model(x_tile, sigma_in.repeat(len(bboxes)), cond=self.repeat_cond_dict(cond_org, bboxes))

        def custom_func(x, bbox_id, bbox):
            tcond = Condition.reconstruct_cond(bbox.cond, step).unsqueeze_(0)
            icond = self.slice_icond(self.get_icond(cond_org), bbox)
            return shared.sd_model.apply_model(x, sigma_in, cond=self.make_cond_dict(cond_in, tcond, icond))

        return self.sample_one_step(x_in, org_func, repeat_func, custom_func)
