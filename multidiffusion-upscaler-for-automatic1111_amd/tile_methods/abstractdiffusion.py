"""
Delegate base class of the tiling methods -- same role, attribute names and method names as upstream
tile_methods/abstractdiffusion.py, so that the Script (scripts/tilediffusion.py) and third-party monkey patches
land on the same surface.  Everything numerical is delegated to the mdtile engine:

    weights / grid init   upstream :24-28, :173-186   -> mdtile.Plan + mdtile_weight_map_add_grid
    custom bbox rects      upstream :194-215            -> host ints here, maps in the subclasses
    reset_buffer           upstream :97-102             -> nothing to clear: the blend is gather-formulated

Out of scope here (host-coupled glue): ControlNet / StableSR tensor slicing and Noise Inversion;
their entry points exist as inert hooks so a caller that pokes them does not crash.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
from torch import Tensor

from modules import devices, shared
from modules.shared import state
from modules.processing import opt_f

import mdtile
from tile_utils.utils import BBox, BBoxSettings, BlendMode, Condition, CustomBBox, Prompt

try:  # isinstance targets; absent in a bare test host
    from modules.sd_samplers_kdiffusion import KDiffusionSampler
except Exception:  # pragma: no cover
    KDiffusionSampler = ()
try:
    from modules.sd_samplers_timesteps import CompVisSampler
except Exception:  # pragma: no cover
    CompVisSampler = ()


class AbstractDiffusion:

    def __init__(self, p, sampler):
        self.method = type(self).__name__
        self.p = p
        self.pbar = None
        self.sampler_name = p.sampler_name
        self.sampler_raw = sampler
        self.sampler = sampler

        if self.is_kdiff and not hasattr(self, "is_edit_model"):
            cfg = self.sampler.model_wrap_cfg
            self.is_edit_model = (shared.sd_model.cond_stage_key == "edit" and cfg.image_cfg_scale is not None
                                  and cfg.image_cfg_scale != 1.0)

        # latent canvas
        self.w: int = int(p.width // opt_f)
        self.h: int = int(p.height // opt_f)
        self.x_buffer: Optional[Tensor] = None   # kept for API compatibility; holds the last blended result
        # sum of tile / background-region weights per latent pixel, fp32 on the device
        self.weights: Tensor = torch.zeros((1, 1, self.h, self.w), device=devices.device, dtype=torch.float32)

        self.step_count = 0
        self.inner_loop_count = 0
        self.kdiff_step = -1

        # grid tiling
        self.enable_grid_bbox = False
        self.plan: Optional[mdtile.Plan] = None
        self.tile_w = self.tile_h = self.tile_bs = self.num_tiles = self.num_batches = None
        self.batched_bboxes: List[List[BBox]] = []

        # region prompt control
        self.enable_custom_bbox = False
        self.custom_bboxes: List[CustomBBox] = []
        self.cond_basis = None
        self.uncond_basis = None
        self.draw_background = True
        self.causal_layers = None

        # inert extension points (ControlNet / StableSR / noise inversion are not part of this engine)
        self.noise_inverse_enabled = False
        self.enable_controlnet = False
        self.enable_stablesr = False

    # ------------------------------------------------------------------------------------------------ host plumbing
    @property
    def is_kdiff(self) -> bool:
        return bool(KDiffusionSampler) and isinstance(self.sampler_raw, KDiffusionSampler)

    @property
    def is_ddim(self) -> bool:
        return bool(CompVisSampler) and isinstance(self.sampler_raw, CompVisSampler)

    def update_pbar(self):
        if self.pbar is None:
            return
        if self.pbar.n >= self.pbar.total:
            self.pbar.close()
        elif self.step_count == state.sampling_step:
            self.inner_loop_count += 1
            if self.inner_loop_count < self.total_bboxes:
                self.pbar.update()
        else:
            self.step_count = state.sampling_step
            self.inner_loop_count = 0

    def reset_buffer(self, x_in: Tensor):
        """Upstream zero-fills an accumulator here (:97-102).  The engine's blend writes every output pixel exactly
        once from a gather, so there is nothing to clear; kept because subclasses / other extensions call it."""
        return None

    def init_done(self):
        self.total_bboxes = 0
        if self.enable_grid_bbox:
            self.total_bboxes += self.num_batches
        if self.enable_custom_bbox:
            self.total_bboxes += len(self.custom_bboxes)
        assert self.total_bboxes > 0, "Nothing to paint! No background to draw and no custom bboxes were provided."
        try:
            from tqdm import tqdm
            self.pbar = tqdm(total=self.total_bboxes * state.sampling_steps, desc=f"{self.method} Sampling: ",
                             disable=getattr(shared.cmd_opts, "mdtile_quiet", False))
        except Exception:  # pragma: no cover
            self.pbar = None

    # ------------------------------------------------------------------------------------------------ cond dict access
    def _tcond_key(self, cond_dict) -> str:
        return "crossattn" if "crossattn" in cond_dict else "c_crossattn"

    def get_tcond(self, cond_dict) -> Tensor:
        t = cond_dict[self._tcond_key(cond_dict)]
        return t[0] if isinstance(t, list) else t

    def set_tcond(self, cond_dict, tcond: Tensor):
        key = self._tcond_key(cond_dict)
        cond_dict[key] = [tcond] if isinstance(cond_dict[key], list) else tcond

    def _icond_key(self, cond_dict) -> str:
        return "c_adm" if shared.sd_model.model.conditioning_key in ("crossattn-adm", "adm") else "c_concat"

    def get_icond(self, cond_dict) -> Tensor:
        i = cond_dict[self._icond_key(cond_dict)]
        return i[0] if isinstance(i, list) else i

    def set_icond(self, cond_dict, icond: Tensor):
        key = self._icond_key(cond_dict)
        cond_dict[key] = [icond] if isinstance(cond_dict[key], list) else icond

    def _vcond_key(self, cond_dict) -> Optional[str]:
        return "vector" if "vector" in cond_dict else None

    def get_vcond(self, cond_dict) -> Optional[Tensor]:
        return cond_dict.get(self._vcond_key(cond_dict))

    def set_vcond(self, cond_dict, vcond: Optional[Tensor]):
        key = self._vcond_key(cond_dict)
        if key is not None:
            cond_dict[key] = vcond

    def make_cond_dict(self, cond_in, tcond: Tensor, icond: Tensor, vcond: Tensor = None):
        out = cond_in.copy()
        self.set_tcond(out, tcond)
        self.set_icond(out, icond)
        self.set_vcond(out, vcond)
        return out

    def slice_icond(self, icond: Tensor, bbox: BBox) -> Tensor:
        """img2img image-conditioning follows the tile (it has the latent's spatial size); txt2img's dummy does not."""
        if tuple(icond.shape[2:]) == (self.h, self.w):
            return icond[bbox.slicer]
        return icond

    # ------------------------------------------------------------------------------------------------ grid
    def get_tile_weights(self):
        """Per-tile weight: scalar 1.0 (MultiDiffusion) or a [tile_h, tile_w] map (Mixture of Diffusers)."""
        return 1.0

    def init_grid_bbox(self, tile_w: int, tile_h: int, overlap: int, tile_bs: int):
        self.enable_grid_bbox = True
        # clamps (tile <= canvas; overlap <= min(requested tile) - 4), origins and batching all happen in the plan
        self.plan = mdtile.Plan(self.w, self.h, tile_w, tile_h, overlap, tile_bs, clamp=True)
        self.tile_w, self.tile_h = self.plan.tile_w, self.plan.tile_h
        self.num_tiles, self.num_batches, self.tile_bs = self.plan.num_tiles, self.plan.num_batches, self.plan.tile_bs
        tile_weights = self.get_tile_weights()
        mdtile.weight_map_add_grid(self.plan, tile_weights if isinstance(tile_weights, Tensor) else None, self.weights)
        self.batched_bboxes = [[BBox(*b) for b in batch] for batch in self.plan.batches]

    # ------------------------------------------------------------------------------------------------ regions
    def init_custom_bbox(self, bbox_settings: Dict[int, BBoxSettings], draw_background: bool, causal_layers: bool):
        self.enable_custom_bbox = True
        self.causal_layers = causal_layers
        self.draw_background = draw_background
        if not draw_background:
            self.enable_grid_bbox = False
            self.weights.zero_()

        self.custom_bboxes = []
        for s in bbox_settings.values():
            enable, fx, fy, fw, fh, prompt, neg, blend_mode, feather_ratio, seed = s
            if not enable or fx > 1.0 or fy > 1.0 or fw <= 0.0 or fh <= 0.0:
                continue
            x, y = max(0, int(fx * self.w)), max(0, int(fy * self.h))
            w, h = min(self.w - x, math.ceil(fw * self.w)), min(self.h - y, math.ceil(fh * self.h))
            self.custom_bboxes.append(CustomBBox(x, y, w, h, prompt, neg, blend_mode, feather_ratio, seed))
        if not self.custom_bboxes:
            self.enable_custom_bbox = False
            return
        self._init_region_conds()

    def _init_region_conds(self):
        """Per-region prompt conditioning via the host's prompt parser (skipped when the host has none, e.g. tests)."""
        p = self.p
        if not hasattr(p, "all_prompts") or getattr(shared.sd_model, "cond_stage_model", None) is None:
            return
        prompts = p.all_prompts[:p.batch_size]
        neg_prompts = p.all_negative_prompts[:p.batch_size]
        for bbox in self.custom_bboxes:
            bbox.cond, bbox.extra_network_data = Condition.get_custom_cond(prompts, bbox.prompt, p.steps, p.styles)
            bbox.uncond = Condition.get_uncond(Prompt.append_prompt(neg_prompts, bbox.neg_prompt), p.steps, p.styles)
        self.cond_basis = Condition.get_cond(prompts, p.steps)
        self.uncond_basis = Condition.get_uncond(neg_prompts, p.steps)

    def blend_plan(self) -> mdtile.Plan:
        """The grid plan, or -- when only custom regions are painted and no grid was initialised -- a one-tile plan that
        merely carries the canvas size for the blend launch."""
        if self.plan is None:
            self.plan = mdtile.Plan(self.w, self.h, self.w, self.h, 0, 1, clamp=True)
        return self.plan

    def region_specs(self, outs: List[Tensor]) -> List[mdtile.RegionSpec]:
        """Pair every custom bbox with its model output for the fused blend launch."""
        specs = []
        for bbox, out, wgt in zip(self.custom_bboxes, outs, self.region_weights()):
            mode = mdtile.REGION_FG if bbox.blend_mode == BlendMode.FOREGROUND else mdtile.REGION_BG
            specs.append(mdtile.RegionSpec(bbox.x, bbox.y, bbox.w, bbox.h, mode, out, wgt))
        return specs

    def region_weights(self) -> List[Optional[Tensor]]:
        """Per region: feather mask (foreground) or the method's background weight (None = 1.0)."""
        return [b.feather_mask if b.blend_mode == BlendMode.FOREGROUND else None for b in self.custom_bboxes]

    # region-prompt forwards: uniform-prompt reconstruction through the host's prompt parser
    def kdiff_custom_forward(self, x_tile: Tensor, sigma_in: Tensor, original_cond, bbox_id: int, bbox: CustomBBox,
                             forward_func):
        """One region evaluated with its own prompt (k-diffusion samplers).  Batched cond+uncond, no 'AND' support:
        the elaborate per-host-version branches of upstream :246-427 are host glue, not part of the engine."""
        step = self.sampler.model_wrap_cfg.step
        tcond = Condition.reconstruct_cond(bbox.cond, step)
        uncond = Condition.reconstruct_uncond(bbox.uncond, step)
        n_cond = x_tile.shape[0] - uncond.shape[0]
        tcond = tcond[:n_cond] if tcond.shape[0] >= n_cond else tcond.expand(n_cond, *tcond.shape[1:])
        icond = self.slice_icond(self.get_icond(original_cond), bbox) if isinstance(original_cond, dict) else None
        cond_in = torch.cat([tcond, uncond], dim=0)
        if isinstance(original_cond, dict):
            cond_out = self.make_cond_dict(original_cond, cond_in, icond, self.get_vcond(original_cond))
        else:
            cond_out = cond_in
        return forward_func(x_tile, sigma_in, cond=cond_out)

    def ddim_custom_forward(self, x: Tensor, cond_in, bbox: CustomBBox, ts: Tensor, forward_func, *args, **kwargs):
        step = getattr(self.sampler, "step", state.sampling_step)
        tcond = Condition.reconstruct_cond(bbox.cond, step)
        uncond = Condition.reconstruct_uncond(bbox.uncond, step)
        icond = self.slice_icond(self.get_icond(cond_in), bbox) if isinstance(cond_in, dict) else None
        if isinstance(cond_in, dict):
            cond = self.make_cond_dict(cond_in, tcond, icond)
            uc = self.make_cond_dict(cond_in, uncond, icond)
        else:
            cond, uc = tcond, uncond
        return forward_func(x, cond, ts, unconditional_conditioning=uc, *args, **kwargs)

    # inert hooks --------------------------------------------------------------------------------------------------------
    def init_controlnet(self, *a, **k): self.enable_controlnet = False
    def init_stablesr(self, *a, **k): self.enable_stablesr = False
    def init_noise_inverse(self, *a, **k): self.noise_inverse_enabled = False
    def reset_controlnet_tensors(self): pass
    def switch_controlnet_tensors(self, *a, **k): pass
    def set_custom_controlnet_tensors(self, *a, **k): pass
    def switch_stablesr_tensors(self, *a, **k): pass
    def set_custom_stablesr_tensors(self, *a, **k): pass
