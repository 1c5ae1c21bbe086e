"""
Delegate base class of the tiling methods -- same role, attribute names and method names as upstream
tile_methods/abstractdiffusion.py, so that the Script (scripts/tilediffusion.py) and third-party monkey patches
land on the same surface.  Everything numerical is delegated to the mdtile engine:

    weights / grid init   upstream :24-28, :173-186   -> mdtile.Plan + mdtile_weight_map_add_grid
    custom bbox rects      upstream :194-215            -> host ints here, maps in the subclasses
    reset_buffer           upstream :97-102             -> nothing to clear: the blend is gather-formulated

    ControlNet / StableSR  upstream :475-588            -> mdtile_gather_rects (slice + cat + repeat in one launch, no tile caches)
    Noise Inversion        upstream :591-747            -> host Euler inversion loop around get_noise (= the blend) and
                                                           mdtile_noise_inverse_blend for the renoise composite (:651-676)
"""
from __future__ import annotations

import math
from types import MethodType
from typing import Dict, List, Optional

import torch
from torch import Tensor

from modules import devices, shared
from modules.shared import state
from modules.processing import opt_f

import mdtile
from tile_utils.utils import BBox, BBoxSettings, BlendMode, Condition, CustomBBox, NoiseInverseCache, Prompt, get_retouch_mask

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

        # optional extensions, armed by the init_* calls of the Script
        self.noise_inverse_enabled = False
        self.sample_img2img_original = None
        self.enable_controlnet = False
        self.controlnet_script = None
        self.control_params = None
        self.org_control_tensor_batch = None
        self.enable_stablesr = False
        self.stablesr_script = None
        self.stablesr_tensor = None

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

    # ---- region-prompt forwards (upstream :232-451) --------------------------------------------------------------------
    def reconstruct_custom_cond(self, org_cond, custom_cond, custom_uncond, bbox: CustomBBox):
        """(region prompt tensor, region negative-prompt tensor, image conditioning cut to the region) at the sampler's step."""
        icond = self.slice_icond(self.get_icond(org_cond), bbox) if isinstance(org_cond, dict) else None
        step = self.sampler.model_wrap_cfg.step
        return Condition.reconstruct_cond(custom_cond, step), Condition.reconstruct_uncond(custom_uncond, step), icond

    def _region_forward(self, forward_func, x, sigma, org_cond, tcond, icond, bbox_id, n_control):
        self.set_custom_controlnet_tensors(bbox_id, n_control)
        self.set_custom_stablesr_tensors(bbox_id)
        cond = self.make_cond_dict(org_cond, tcond, icond, self.get_vcond(org_cond)) if isinstance(org_cond, dict) else tcond
        return forward_func(x, sigma, cond=cond)

    def _split_forward(self, forward_func, x_tile, sigma_in, org_cond, first, second, icond, bbox_id, tail_repeats_second=False):
        """Two model calls for prompt tensors of different token length (they cannot be concatenated); rows beyond the two chunks
        (edit models: a second uncond copy) reuse the second result."""
        n1, n2 = first.shape[0], second.shape[0]
        x_out = torch.zeros_like(x_tile)
        ic1 = icond[:n1] if icond is not None and icond.shape[0] >= n1 + n2 else icond
        ic2 = icond[n1:n1 + n2] if icond is not None and icond.shape[0] >= n1 + n2 else icond
        x_out[:n1] = self._region_forward(forward_func, x_tile[:n1], sigma_in[:n1], org_cond, first, ic1, bbox_id, n1)
        out2 = self._region_forward(forward_func, x_tile[n1:n1 + n2], sigma_in[n1:n1 + n2], org_cond, second, ic2, bbox_id, n2)
        x_out[n1:n1 + n2] = out2
        if tail_repeats_second and x_tile.shape[0] > n1 + n2:
            x_out[n1 + n2:] = out2
        return x_out

    def kdiff_custom_forward(self, x_tile: Tensor, sigma_in: Tensor, original_cond, bbox_id: int, bbox: CustomBBox,
                             forward_func):
        """One region evaluated with its own prompt under a k-diffusion sampler.  The host's CFG denoiser feeds either the whole
        [cond, uncond(, uncond)] batch at once or slices of it (batch_cond_uncond off, --lowvram / --medvram, prompts of different
        token length); the region's tensors are sliced the same way (upstream :246-427)."""
        step = self.sampler.model_wrap_cfg.step
        if self.kdiff_step != step:               # first region call of a sampler step: forget the per-step progress
            self.kdiff_step = step
            self.kdiff_step_bbox = [-1] * len(self.custom_bboxes)
            self.tensor, self.uncond, self.image_cond_in = {}, {}, {}
            # the GLOBAL prompts only tell how the host batches this step
            self.real_tensor = Condition.reconstruct_cond(self.cond_basis, step)
            self.real_uncond = Condition.reconstruct_uncond(self.uncond_basis, step)
            self.a = [0] * len(self.custom_bboxes)
        same_len_global = self.real_tensor.shape[1] == self.real_uncond.shape[1]
        edit = bool(getattr(self, "is_edit_model", False))

        if self.kdiff_step_bbox[bbox_id] != step:
            self.kdiff_step_bbox[bbox_id] = step
            tensor, uncond, icond = self.reconstruct_custom_cond(original_cond, bbox.cond, bbox.uncond, bbox)
            if same_len_global and shared.batch_cond_uncond:
                # the whole batch is in x_tile
                if tensor.shape[1] == uncond.shape[1]:
                    cond = torch.cat([tensor, uncond, uncond] if edit else [tensor, uncond])
                    return self._region_forward(forward_func, x_tile, sigma_in, original_cond, cond, icond, bbox_id, x_tile.shape[0])
                return self._split_forward(forward_func, x_tile, sigma_in, original_cond, tensor, uncond, icond, bbox_id, tail_repeats_second=edit)
            self.tensor[bbox_id], self.uncond[bbox_id], self.image_cond_in[bbox_id] = tensor, uncond, icond

        # partial batches: rows [a, b) of the virtual [tensor, uncond(, uncond)] stack
        tensor, uncond, icond = self.tensor[bbox_id], self.uncond[bbox_id], self.image_cond_in[bbox_id]
        a = self.a[bbox_id]
        b = a + x_tile.shape[0]
        self.a[bbox_id] = b
        nt, nu = tensor.shape[0], uncond.shape[0]
        if same_len_global:
            # segments of the stack covered by [a, b): (source, lo, hi)
            stack = [(tensor, 0, nt), (uncond, nt, nt + nu)] + ([(uncond, nt + nu, nt + 2 * nu)] if edit else [])
            parts = [src[max(a, lo) - lo:min(b, hi) - lo] for src, lo, hi in stack if max(a, lo) < min(b, hi)]
            if len(parts) == 1 or all(p_.shape[1] == parts[0].shape[1] for p_ in parts):
                cond = parts[0] if len(parts) == 1 else torch.cat(parts)
                return self._region_forward(forward_func, x_tile, sigma_in, original_cond, cond, icond, bbox_id, x_tile.shape[0])
            first, second = parts[0], torch.cat(parts[1:])
            return self._split_forward(forward_func, x_tile, sigma_in, original_cond, first, second, icond, bbox_id)
        # global prompts of different length: the host runs cond and uncond in separate calls, so do the regions
        if a < nt:
            cond = tensor[a:b] if not edit else torch.cat([tensor[a:b], uncond])
            return self._region_forward(forward_func, x_tile, sigma_in, original_cond, cond, icond, bbox_id, x_tile.shape[0])
        return self._region_forward(forward_func, x_tile, sigma_in, original_cond, uncond, icond, bbox_id, uncond.shape[0])

    def ddim_custom_forward(self, x: Tensor, cond_in, bbox: CustomBBox, ts: Tensor, forward_func, *args, **kwargs):
        """One region under a CompVis (DDIM / PLMS) sampler: cond and uncond travel as two arguments and are concatenated later,
        so the negative prompt is padded with its last token vector / truncated to the prompt's length (upstream :429-451)."""
        tcond, uncond, icond = self.reconstruct_custom_cond(cond_in, bbox.cond, bbox.uncond, bbox)
        if uncond.shape[1] < tcond.shape[1]:
            uncond = torch.hstack([uncond, uncond[:, -1:].repeat([1, tcond.shape[1] - uncond.shape[1], 1])])
        elif uncond.shape[1] > tcond.shape[1]:
            uncond = uncond[:, :tcond.shape[1]]
        if icond is not None:
            cond, uc = self.make_cond_dict(cond_in, tcond, icond), self.make_cond_dict(cond_in, uncond, icond)
        else:
            cond, uc = tcond, uncond
        return forward_func(x, cond, ts, unconditional_conditioning=uc, *args, **kwargs)

    # ---- ControlNet (upstream :454-544) ---------------------------------------------------------------------------------
    # Upstream crops every hint into per-batch tile stacks at init (optionally parked on the CPU) and re-assembles / repeats /
    # uploads them at every model call.  Here the full hints stay where they are and one mdtile_gather_rects launch per hint
    # produces the sliced + repeated tensor when a batch is switched in -- nothing is cached.
    def init_controlnet(self, controlnet_script, control_tensor_cpu: bool):
        self.enable_controlnet = True
        self.controlnet_script = controlnet_script
        self.control_tensor_cpu = control_tensor_cpu      # accepted for the UI's sake: there are no tile caches to park
        self.control_params = None
        self.org_control_tensor_batch = None
        self.prepare_controlnet_tensors()

    def reset_controlnet_tensors(self):
        if not self.enable_controlnet or self.org_control_tensor_batch is None:
            return
        for param, hint in zip(self.control_params, self.org_control_tensor_batch):
            param.hint_cond = hint

    def prepare_controlnet_tensors(self, refresh: bool = False):
        """Remember the network's control params and their ORIGINAL hints (full canvas, opt_f x the latent grid)."""
        if not refresh and self.control_params is not None:
            return
        if not self.enable_controlnet or self.controlnet_script is None:
            return
        net = getattr(self.controlnet_script, "latest_network", None)
        if net is None or not hasattr(net, "control_params"):
            return
        self.control_params = net.control_params
        hints = []
        for param in self.control_params:
            h = param.hint_cond
            hints.append(h.unsqueeze(0) if h.dim() == 3 else h)
        self.org_control_tensor_batch = hints

    def _hint_on_device(self, hint: Tensor) -> Tensor:
        return hint if hint.device.type == "cuda" else hint.to(devices.device)

    def switch_controlnet_tensors(self, batch_id: int, x_batch_size: int, tile_batch_size: int, is_denoise: bool = False):
        if not self.enable_controlnet or not self.org_control_tensor_batch:
            return
        bboxes = self.batched_bboxes[batch_id]
        rects = [(b.x * opt_f, b.y * opt_f) for b in bboxes]
        w, h = bboxes[0].w * opt_f, bboxes[0].h * opt_f
        for param, hint in zip(self.control_params, self.org_control_tensor_batch):
            hint = self._hint_on_device(hint).contiguous()
            if self.is_kdiff:      # every tile's x_batch_size copies are consecutive (tile-major model batch)
                param.hint_cond = mdtile.gather_rects(hint[:1], rects[:tile_batch_size], w, h, repeat=x_batch_size, tile_major=True)
            else:                  # DDIM: the tile stack as a whole, repeated for cond + uncond (once when only denoising)
                param.hint_cond = mdtile.gather_rects(hint, rects, w, h, repeat=x_batch_size if is_denoise else x_batch_size * 2, tile_major=False)

    def set_custom_controlnet_tensors(self, bbox_id: int, repeat_size: int):
        if not self.enable_controlnet or not self.org_control_tensor_batch or not self.custom_bboxes:
            return
        bbox = self.custom_bboxes[bbox_id]
        for param, hint in zip(self.control_params, self.org_control_tensor_batch):
            hint = self._hint_on_device(hint).contiguous()
            param.hint_cond = mdtile.gather_rects(hint, [(bbox.x * opt_f, bbox.y * opt_f)], bbox.w * opt_f, bbox.h * opt_f,
                                                  repeat=repeat_size, tile_major=False)

    # ---- StableSR (upstream :547-588) --------------------------------------------------------------------------------------
    def init_stablesr(self, stablesr_script):
        if stablesr_script.stablesr_model is None:
            return
        self.stablesr_script = stablesr_script

        def set_image_hook(latent_image: Tensor):
            self.enable_stablesr = True
            self.stablesr_tensor = latent_image

        stablesr_script.stablesr_model.set_image_hooks["TiledDiffusion"] = set_image_hook

    def _stablesr_live(self) -> bool:
        return self.enable_stablesr and self.stablesr_script is not None and self.stablesr_script.stablesr_model is not None \
            and self.stablesr_tensor is not None

    def reset_stablesr_tensors(self):
        if self._stablesr_live():
            self.stablesr_script.stablesr_model.latent_image = self.stablesr_tensor

    def switch_stablesr_tensors(self, batch_id: int):
        if not self._stablesr_live():
            return
        bboxes = self.batched_bboxes[batch_id]
        t = self.stablesr_tensor.contiguous()
        self.stablesr_script.stablesr_model.latent_image = mdtile.gather_rects(t, [(b.x, b.y) for b in bboxes], bboxes[0].w, bboxes[0].h)

    def set_custom_stablesr_tensors(self, bbox_id: int):
        if not self._stablesr_live() or not self.custom_bboxes:
            return
        bbox = self.custom_bboxes[bbox_id]
        self.stablesr_script.stablesr_model.latent_image = mdtile.gather_rect(self.stablesr_tensor.contiguous(), bbox.x, bbox.y, bbox.w, bbox.h)

    # ---- Noise Inversion (upstream :591-747) -----------------------------------------------------------------------------
    def init_noise_inverse(self, steps: int, retouch: float, get_cache_callback, set_cache_callback, renoise_strength: float,
                           renoise_kernel: int):
        self.noise_inverse_enabled = True
        self.noise_inverse_steps = steps
        self.noise_inverse_retouch = float(retouch)
        self.noise_inverse_renoise_strength = float(renoise_strength)
        self.noise_inverse_renoise_kernel = int(renoise_kernel)
        if self.sample_img2img_original is None:
            self.sample_img2img_original = self.sampler_raw.sample_img2img
        self.sampler_raw.sample_img2img = MethodType(self.sample_img2img, self.sampler_raw)
        self.noise_inverse_set_cache = set_cache_callback
        self.noise_inverse_get_cache = get_cache_callback

    def renoise_mask(self, p, size) -> Optional[Tensor]:
        """[H, W] weight of fresh noise: where the guided filter says the input image has detail, scaled by the renoise strength
        (upstream :607-616)."""
        if self.noise_inverse_renoise_strength <= 0:
            return None
        import numpy as np
        import torch.nn.functional as F
        gray = np.asarray(p.init_images[0].convert("L"))
        m = torch.from_numpy(get_retouch_mask(gray, self.noise_inverse_renoise_kernel)).to(devices.device)
        m = 1 - F.interpolate(m.unsqueeze(0).unsqueeze(0), size=size, mode="bilinear").squeeze(0).squeeze(0)
        m *= self.noise_inverse_renoise_strength
        return torch.clamp(m, 0, 1)

    def _cached_inversion(self, p, prompts) -> Optional[Tensor]:
        c = self.noise_inverse_get_cache()
        if c is None:
            return None
        same = (c.model_hash == p.sd_model.sd_model_hash and c.noise_inversion_steps == self.noise_inverse_steps
                and len(c.prompts) == len(prompts) and all(a == b for a, b in zip(c.prompts, prompts))
                and abs(c.retouch - self.noise_inverse_retouch) < 0.01 and c.x0.shape == p.init_latent.shape
                and torch.abs(c.x0.to(p.init_latent.device) - p.init_latent).sum() < 100)
        return c.xt if same else None

    def sample_img2img(self, sampler, p, x: Tensor, noise: Tensor, conditioning, unconditional_conditioning, steps=None,
                       image_conditioning=None):
        """Replacement of the sampler's sample_img2img: the img2img start noise is the noise that INVERTS the init image
        (tiled Euler inversion through get_noise), mixed with fresh noise where the image has detail (upstream :606-681)."""
        from modules import sd_samplers_common
        renoise_mask = self.renoise_mask(p, tuple(noise.shape[-2:]))
        prompts = p.all_prompts[:p.batch_size]
        latent = self._cached_inversion(p, prompts)
        if latent is not None:
            print("[Tiled Diffusion] checkpoint, image, prompts, inversion steps and retouch are unchanged: "
                  "Noise Inversion reuses the latent of the previous run.")
            latent = latent.to(noise.device)
        else:
            shared.state.job_count += 1
            latent = self.find_noise_for_image_sigma_adjustment(sampler.model_wrap, self.noise_inverse_steps, prompts)
            shared.state.nextjob()
            self.noise_inverse_set_cache(p.init_latent.clone().cpu(), latent.clone().cpu(), prompts)
        adjusted_steps, _ = sd_samplers_common.setup_img2img_steps(p, steps)
        sigmas = sampler.get_sigmas(p, adjusted_steps)
        inverse_noise = latent - (p.init_latent / sigmas[0])
        if renoise_mask is not None:
            # :655-676 -- one engine launch.  With the grid disabled the job's noise is first re-weighted by the regions.
            regions = []
            if not self.enable_grid_bbox:
                for b in self.custom_bboxes:
                    fg = b.blend_mode == BlendMode.FOREGROUND
                    regions.append((b.x, b.y, b.w, b.h, mdtile.REGION_FG if fg else mdtile.REGION_BG,
                                    b.feather_mask.to(device=noise.device, dtype=torch.float32).contiguous() if fg else None))
            combined = mdtile.noise_inverse_blend(noise.float().contiguous(), inverse_noise.float().contiguous(),
                                                  renoise_mask.float().contiguous(), regions).to(noise.dtype)
        else:
            combined = inverse_noise
        return self.sample_img2img_original(p, x, combined, conditioning, unconditional_conditioning, steps, image_conditioning)

    @torch.no_grad()
    def find_noise_for_image_sigma_adjustment(self, dnw, steps: int, prompts: List[str]) -> Tensor:
        """Euler inversion of the init latent over `steps` sigmas, every model call tiled through get_noise (upstream :683-743,
        after the host's img2imgalt script)."""
        import k_diffusion as K
        from modules import sd_samplers_common
        assert self.p.sampler_name == "Euler"
        x = self.p.init_latent
        s_in = x.new_ones([x.shape[0]])
        skip = 1 if shared.sd_model.parameterization == "v" else 0
        sigmas = dnw.get_sigmas(steps).flip(0)
        cond = self.p.sd_model.get_learned_conditioning(prompts)
        if isinstance(cond, Tensor):     # SD1 / SD2
            cond_in = self.make_cond_dict({"c_crossattn": [], "c_concat": []}, cond, self.p.image_conditioning)
        else:                            # SDXL
            cond_in = self.make_cond_dict({"crossattn": None, "vector": None, "c_concat": []}, cond["crossattn"],
                                          self.p.image_conditioning, cond["vector"])
        state.sampling_steps = steps
        try:
            from tqdm import tqdm
            bar = tqdm(total=steps, desc="Noise Inversion")
        except Exception:  # pragma: no cover
            bar = None
        for i in range(1, len(sigmas)):
            if state.interrupted:
                return x
            state.sampling_step += 1
            sigma_in = torch.cat([sigmas[i] * s_in])
            c_out, c_in = [K.utils.append_dims(k, x.ndim) for k in dnw.get_scalings(sigma_in)[skip:]]
            t = dnw.sigma_to_t(sigma_in) / self.noise_inverse_retouch
            eps = self.get_noise(x * c_in, t, cond_in, steps - i)
            denoised = x + eps * c_out
            d = (x - denoised) / sigmas[i]            # Euler step towards the next (larger) sigma
            x = x + d * (sigmas[i] - sigmas[i - 1])
            sd_samplers_common.store_latent(x)
            del sigma_in, c_out, c_in, t, eps, denoised, d
            if bar is not None:
                bar.update(1)
        if bar is not None:
            bar.close()
        return x / sigmas[-1]

    def get_noise(self, x_in: Tensor, sigma_in: Tensor, cond_in, step: int) -> Tensor:
        raise NotImplementedError
