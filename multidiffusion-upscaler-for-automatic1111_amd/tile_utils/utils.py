"""
Tile utilities of the plugin surface (same public names as upstream tile_utils/utils.py), with every map / grid
computation routed through the mdtile engine (libmdtile.so, include/mdtile.h):

    split_bboxes      upstream utils.py:160-177  -> mdtile_plan_create + mdtile_weight_map_add_grid
    gaussian_weights  upstream utils.py:180-194  -> mdtile_gaussian_weights
    feather_mask      upstream utils.py:196-214  -> mdtile_feather_mask

Prompt / cond helpers stay thin host-side Python (they only forward to `modules.prompt_parser`).
`get_retouch_mask` (cv2 guided filter, once per Noise Inversion job on the CPU) is host glue as upstream.
"""
from __future__ import annotations

import math
from collections import namedtuple
from enum import Enum
from typing import Any, Dict, List, Tuple, Union

import torch
from torch import Tensor

from modules import devices, shared
from modules.processing import opt_f

import mdtile


class ComparableEnum(Enum):
    def __eq__(self, other: Any) -> bool:
        if isinstance(other, str):
            return self.value == other
        if isinstance(other, ComparableEnum):
            return self.value == other.value
        raise TypeError(f"unsupported type: {type(other)}")

    __hash__ = Enum.__hash__


class Method(ComparableEnum):
    MULTI_DIFF = "MultiDiffusion"
    MIX_DIFF = "Mixture of Diffusers"


class Method_2(ComparableEnum):
    DEMO_FU = "DemoFusion"


class BlendMode(Enum):
    FOREGROUND = "Foreground"
    BACKGROUND = "Background"


# field order == the 10 gradio controls of one region block == upstream BBoxSettings (utils.py:41)
BBoxSettings = namedtuple("BBoxSettings", ["enable", "x", "y", "w", "h", "prompt", "neg_prompt", "blend_mode",
                                           "feather_ratio", "seed"])
NoiseInverseCache = namedtuple("NoiseInversionCache", ["model_hash", "x0", "xt", "noise_inversion_steps", "retouch", "prompts"])
DEFAULT_BBOX_SETTINGS = BBoxSettings(False, 0.4, 0.4, 0.2, 0.2, "", "", BlendMode.BACKGROUND.value, 0.2, -1)
NUM_BBOX_PARAMS = len(BBoxSettings._fields)


def build_bbox_settings(bbox_control_states: List[Any]) -> Dict[int, BBoxSettings]:
    """Positional gradio values -> {region index: settings}; floats rounded to 4 digits, disabled / degenerate regions
    dropped (upstream utils.py:47-63)."""
    out: Dict[int, BBoxSettings] = {}
    for index, start in enumerate(range(0, len(bbox_control_states), NUM_BBOX_PARAMS)):
        s = BBoxSettings(*bbox_control_states[start:start + NUM_BBOX_PARAMS])
        s = s._replace(x=round(s.x, 4), y=round(s.y, 4), w=round(s.w, 4), h=round(s.h, 4),
                       feather_ratio=round(s.feather_ratio, 4), seed=int(s.seed))
        if s.enable and s.x <= 1.0 and s.y <= 1.0 and s.w > 0.0 and s.h > 0.0:
            out[index] = s
    return out


def gr_value(value=None, visible=None):
    return {"value": value, "visible": visible, "__type__": "update"}


class BBox:
    """Grid tile rectangle in latent pixels; `slicer` indexes an NCHW tensor."""

    def __init__(self, x: int, y: int, w: int, h: int):
        self.x, self.y, self.w, self.h = x, y, w, h
        self.box = [x, y, x + w, y + h]
        self.slicer = (slice(None), slice(None), slice(y, y + h), slice(x, x + w))

    def __getitem__(self, idx: int) -> int:
        return self.box[idx]

    def __repr__(self):
        return f"{type(self).__name__}(x={self.x}, y={self.y}, w={self.w}, h={self.h})"


class CustomBBox(BBox):
    """Region-prompt rectangle.  Foreground regions carry their feather mask (built on the GPU by the engine)."""

    def __init__(self, x: int, y: int, w: int, h: int, prompt: str, neg_prompt: str, blend_mode: str,
                 feather_radio: float, seed: int):
        super().__init__(x, y, w, h)
        self.prompt, self.neg_prompt = prompt, neg_prompt
        self.blend_mode = BlendMode(blend_mode)
        self.feather_ratio = max(min(feather_radio, 1.0), 0.0)
        self.seed = seed
        self.feather_mask = feather_mask(w, h, self.feather_ratio) if self.blend_mode == BlendMode.FOREGROUND else None
        self.cond = None
        self.extra_network_data = None
        self.uncond = None


class Prompt:
    @staticmethod
    def apply_styles(prompts: List[str], styles=None) -> List[str]:
        if not styles:
            return prompts
        return [shared.prompt_styles.apply_styles_to_prompt(p, styles) for p in prompts]

    @staticmethod
    def append_prompt(prompts: List[str], prompt: str = "") -> List[str]:
        if not prompt:
            return prompts
        return [f"{p}, {prompt}" for p in prompts]


class Condition:
    """Thin forwards to the host's prompt parser (host-coupled, no arithmetic)."""

    @staticmethod
    def get_custom_cond(prompts: List[str], prompt, steps: int, styles=None):
        from modules import extra_networks
        prompt = Prompt.apply_styles([prompt], styles)[0]
        _, extra_network_data = extra_networks.parse_prompts([prompt])
        prompts = Prompt.apply_styles(Prompt.append_prompt(prompts, prompt), styles)
        return Condition.get_cond(prompts, steps), extra_network_data

    @staticmethod
    def get_cond(prompts, steps: int):
        from modules import extra_networks, prompt_parser
        prompts, _ = extra_networks.parse_prompts(prompts)
        return prompt_parser.get_multicond_learned_conditioning(shared.sd_model, prompts, steps)

    @staticmethod
    def get_uncond(neg_prompts: List[str], steps: int, styles=None):
        from modules import prompt_parser
        return prompt_parser.get_learned_conditioning(shared.sd_model, Prompt.apply_styles(neg_prompts, styles), steps)

    @staticmethod
    def reconstruct_cond(cond, step: int) -> Tensor:
        from modules import prompt_parser
        _, tensor = prompt_parser.reconstruct_multicond_batch(cond, step)
        return tensor

    @staticmethod
    def reconstruct_uncond(uncond, step: int) -> Tensor:
        from modules import prompt_parser
        return prompt_parser.reconstruct_cond_batch(uncond, step)


def splitable(w: int, h: int, tile_w: int, tile_h: int, overlap: int = 16) -> bool:
    """More than one tile for an IMAGE-space canvas (w, h)?  (upstream utils.py:151-158)"""
    w, h = w // opt_f, h // opt_f
    m = min(tile_w, tile_h)
    if overlap >= m:
        overlap = m - 4
    return math.ceil((w - overlap) / (tile_w - overlap)) > 1 or math.ceil((h - overlap) / (tile_h - overlap)) > 1


def split_bboxes(w: int, h: int, tile_w: int, tile_h: int, overlap: int = 16,
                 init_weight: Union[Tensor, float] = 1.0) -> Tuple[List[BBox], Tensor]:
    """Overlapping tile grid + its summed weight map, computed by the engine (raw grid, no clamping)."""
    plan = mdtile.Plan(w, h, tile_w, tile_h, overlap, 1, clamp=False)
    weight = torch.zeros((1, 1, h, w), device=devices.device, dtype=torch.float32)
    tile_w_map = None
    if isinstance(init_weight, Tensor):
        tile_w_map = init_weight.to(device=devices.device, dtype=torch.float32).contiguous()
        mdtile.weight_map_add_grid(plan, tile_w_map, weight)
    else:
        mdtile.weight_map_add_grid(plan, None, weight)
        if float(init_weight) != 1.0:
            weight *= float(init_weight)
    return [BBox(*b) for b in plan.bboxes], weight


def gaussian_weights(tile_w: int, tile_h: int) -> Tensor:
    """Mixture-of-Diffusers tile weight (upstream utils.py:180-194), generated on the GPU in fp64 -> fp32."""
    return mdtile.gaussian_weights(tile_w, tile_h, devices.device)


def feather_mask(w: int, h: int, ratio: float) -> Tensor:
    """Foreground feather mask (upstream utils.py:196-214), generated on the GPU."""
    return mdtile.feather_mask(w, h, ratio, devices.device)


NoiseInverseCache = namedtuple("NoiseInversionCache", ["model_hash", "x0", "xt", "noise_inversion_steps", "retouch", "prompts"])


def get_retouch_mask(img_input, kernel_size: int):
    """Where a grey image [H, W] uint8 carries detail: the residue of a self-guided box filter (guided filter with guide = input,
    eps 0.01), as uint8-quantised fractions in [0, 1] float32 (upstream tile_utils/utils.py:216-247).  Host glue of Noise
    Inversion: needs OpenCV, runs once per job on the CPU."""
    import cv2
    import numpy as np
    k = (int(round(kernel_size)), int(round(kernel_size)))
    img = img_input.astype(np.float32) / 255.0
    mean = cv2.blur(img, k)
    var = cv2.blur(img * img, k) - mean * mean
    a = var / (var + 0.01)                 # cov(I, I) / (var(I) + eps)
    b = mean - a * mean
    gf = (a * img + b) - img
    gf *= 255
    gf = gf.astype(np.uint8)               # upstream quantises (and wraps negatives) exactly like this
    return gf.clip(0, 255).astype(np.float32) / 255.0


def null_decorator(fn):
    return fn


keep_signature = controlnet = stablesr = grid_bbox = custom_bbox = noise_inverse = null_decorator
