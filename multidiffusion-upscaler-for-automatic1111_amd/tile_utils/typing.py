"""Host type aliases (counterpart of upstream tile_utils/typing.py:1-44): everything is optional at import time so the
engine and its tests also run without a webui; under A1111 the real classes are picked up."""
from typing import Any, Callable, DefaultDict, Dict, List, Optional, Tuple, Union  # noqa: F401 (re-exported)

from torch import Tensor  # noqa: F401

NoType = Any


def _opt(module: str, *names: str):
    try:
        mod = __import__(module, fromlist=list(names))
        return tuple(getattr(mod, n, NoType) for n in names)
    except Exception:  # host absent or too old: degrade to Any
        return tuple(NoType for _ in names)


(Processing, ProcessingImg2Img, Processed) = _opt(
    "modules.processing", "StableDiffusionProcessing", "StableDiffusionProcessingImg2Img", "Processed")
(MulticondLearnedConditioning, ScheduledPromptConditioning) = _opt(
    "modules.prompt_parser", "MulticondLearnedConditioning", "ScheduledPromptConditioning")
(ExtraNetworkParams,) = _opt("modules.extra_networks", "ExtraNetworkParams")
(KDiffusionSampler, CFGDenoiser, CFGDenoiserKDiffusion) = _opt(
    "modules.sd_samplers_kdiffusion", "KDiffusionSampler", "CFGDenoiser", "CFGDenoiserKDiffusion")
(CompVisSampler, CFGDenoiserTimesteps, CompVisTimestepsDenoiser, CompVisTimestepsVDenoiser) = _opt(
    "modules.sd_samplers_timesteps", "CompVisSampler", "CFGDenoiserTimesteps", "CompVisTimestepsDenoiser",
    "CompVisTimestepsVDenoiser")
(CompVisDenoiser, CompVisVDenoiser) = _opt("k_diffusion.external", "CompVisDenoiser", "CompVisVDenoiser")
(LatentDiffusion,) = _opt("ldm.models.diffusion.ddpm", "LatentDiffusion")
(State,) = _opt("modules.shared_state", "State")
if State is NoType:
    (State,) = _opt("modules.shared", "State")

Sampler = Any
Cond = Any
Uncond = Any
ExtraNetworkData = Any
# 'c_crossattn' [B,77,768] / 'crossattn' (SDXL) = text cond; 'c_concat' [B,5,H,W] = image cond; 'vector' (SDXL)
CondDict = Dict[str, Any]
