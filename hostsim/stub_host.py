"""
TEST INFRASTRUCTURE -- not part of the shipped product path.

A minimal stand-in for the A1111 (stable-diffusion-webui) host so that
  (1) the upstream reference at /root/reference can be imported *verbatim* on CPU
      (only in the build container -- the GPU box has no /root/reference), and
  (2) this repo's own plugin (multidiffusion-upscaler-for-automatic1111_amd/) can be
      imported and driven by tests, bench.py and __graft_entry__.smoke() without a webui.

The symbol list follows SURVEY.md Appendix B/E: every name the reference touches at import time
(`tile_utils/utils.py:1-16`, `tile_utils/typing.py:1-30`, `scripts/tilevae.py:52-76`,
`tile_utils/attn.py:5-16`) exists here with the weakest behaviour that lets the hot path run.
Nothing in here computes anything on the hot path.
"""
from __future__ import annotations

import contextlib
import importlib
import os
import sys
import types
from types import SimpleNamespace

import torch

REFERENCE_ROOT = os.environ.get("MDTILE_REFERENCE_ROOT", "/root/reference")
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN_ROOT = os.path.join(REPO_ROOT, "multidiffusion-upscaler-for-automatic1111_amd")

_INSTALLED = False


class _State:
    """modules.shared_state.State look-alike (only the attributes the tiling code polls)."""

    def __init__(self):
        self.interrupted = False
        self.skipped = False
        self.sampling_step = 0
        self.sampling_steps = 20
        self.job_count = 0
        self.job_no = 0

    def nextjob(self):
        self.job_no += 1


class NansException(Exception):
    pass


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def _test_for_nans(x, where):
    if torch.isnan(x).any().item():
        raise NansException(f"A tensor with NaNs was produced in {where}.")


def install(device: str | torch.device | None = None) -> types.ModuleType:
    """Register the fake host modules in sys.modules (idempotent). Returns `modules.shared`.
    `device=None` keeps the current `devices.device` of an already-installed stub (cpu on first install)."""
    global _INSTALLED
    if _INSTALLED:
        if device is not None:
            set_device(device)
        return sys.modules["modules.shared"]
    dev = torch.device(device if device is not None else "cpu")

    sys.dont_write_bytecode = True  # /root/reference is a read-only mount

    class Script:  # modules.scripts.Script
        def title(self):
            raise NotImplementedError

        def show(self, is_img2img):
            return True

        def ui(self, is_img2img):
            return []

    class _Denoiser:
        def forward(self, *a, **k):
            raise NotImplementedError

    class KDiffusionSampler:  # isinstance() is used on these (abstractdiffusion.py:77-83)
        def __init__(self):
            # what the delegates read from the CFG denoiser wrapper (abstractdiffusion.py:17-20, 240)
            self.model_wrap_cfg = SimpleNamespace(image_cfg_scale=None, step=0, inner_model=_Denoiser())

    class CompVisSampler:
        pass

    class LatentDiffusion:
        def apply_model(self, *a, **k):
            raise NotImplementedError

    class _Any:
        def __init__(self, *a, **k):
            pass

    _mod("modules")
    _mod(
        "modules.devices",
        device=dev,
        cpu=torch.device("cpu"),
        torch_gc=lambda: None,
        test_for_nans=_test_for_nans,
        autocast=contextlib.nullcontext,
        get_optimal_device=lambda: sys.modules["modules.devices"].device,
        get_optimal_device_name=lambda: sys.modules["modules.devices"].device.type,
        NansException=NansException,
    )
    state = _State()
    opts = SimpleNamespace(upcast_attn=False, img2img_background_color="#ffffff")
    cmd_opts = SimpleNamespace()
    _mod("modules.shared_state", State=_State)
    shared = _mod(
        "modules.shared",
        state=state,
        opts=opts,
        cmd_opts=cmd_opts,
        sd_model=SimpleNamespace(model=SimpleNamespace(conditioning_key="crossattn"), cond_stage_key="txt"),
        batch_cond_uncond=True,
        State=_State,
        sd_upscalers=[],
        prompt_styles=None,
    )
    _mod("modules.prompt_parser", MulticondLearnedConditioning=_Any, ScheduledPromptConditioning=_Any)
    _mod(
        "modules.extra_networks",
        ExtraNetworkParams=_Any,
        activate=lambda p, data: None,
        deactivate=lambda p, data: None,
        parse_prompts=lambda prompts: (prompts, {}),
    )
    _mod("modules.sd_samplers_common", InterruptedException=type("InterruptedException", (BaseException,), {}),
         Sampler=type("Sampler", (), {"callback_state": lambda self, d: None}),
         setup_img2img_steps=lambda p, steps=None: (steps or getattr(p, "steps", 20), (steps or getattr(p, "steps", 20)) - 1),
         store_latent=lambda x: None)
    _mod("modules.images", resize_image=lambda *a, **k: None)
    _mod("modules.sd_samplers", create_sampler=lambda name, model: KDiffusionSampler())
    _mod(
        "modules.processing",
        opt_f=8,
        StableDiffusionProcessing=_Any,
        StableDiffusionProcessingImg2Img=_Any,
        Processed=_Any,
        create_random_tensors=None,
        get_fixed_seed=lambda seed: int(seed) if seed not in (None, "", -1) else 1234567,
    )
    _mod("modules.sd_samplers_kdiffusion", KDiffusionSampler=KDiffusionSampler, CFGDenoiser=_Any,
         CFGDenoiserKDiffusion=_Any)
    _mod("modules.sd_samplers_timesteps", CompVisSampler=CompVisSampler, CFGDenoiserTimesteps=_Any,
         CompVisTimestepsDenoiser=_Denoiser, CompVisTimestepsVDenoiser=_Denoiser)
    _mod("modules.scripts", Script=Script, AlwaysVisible=object(), basedir=lambda: PLUGIN_ROOT)
    _mod("modules.ui", gr_show=lambda visible=True: {"visible": visible, "__type__": "update"})
    _mod("modules.sd_vae_approx", cheap_approximation=lambda x: x[:3])
    _mod("modules.sd_hijack", model_hijack=SimpleNamespace(optimization_method=None))
    _mod("modules.sd_hijack_optimizations", get_available_vram=lambda: 0,
         get_xformers_flash_attention_op=lambda *a: None, sub_quad_attention=None)
    _mod("cv2")
    _mod("gradio")
    _mod("gradio.components", Component=_Any)
    _mod("k_diffusion")
    _mod("k_diffusion.external", CompVisDenoiser=_Denoiser, CompVisVDenoiser=_Denoiser)
    _mod("ldm")
    _mod("ldm.models")
    _mod("ldm.models.diffusion")
    _mod("ldm.models.diffusion.ddpm", LatentDiffusion=LatentDiffusion)
    _mod("ldm.modules")
    _mod("ldm.modules.diffusionmodules")
    _mod("ldm.modules.diffusionmodules.model", AttnBlock=_Any, MemoryEfficientAttnBlock=_Any)
    _INSTALLED = True
    return shared


def set_device(device) -> None:
    sys.modules["modules.devices"].device = torch.device(device)


def host():
    """(devices, shared) modules of the installed stub."""
    return sys.modules["modules.devices"], sys.modules["modules.shared"]


_PLUGIN_TOPLEVEL = ("tile_utils", "tile_methods", "scripts")


def _evict(prefixes) -> dict:
    saved = {}
    for name in list(sys.modules):
        if any(name == p or name.startswith(p + ".") for p in prefixes):
            saved[name] = sys.modules.pop(name)
    return saved


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "tile_utils", "utils.py"))


_REF_CACHE = None


def load_reference() -> SimpleNamespace:
    """Import the upstream modules verbatim from REFERENCE_ROOT and return them in a namespace.

    The reference and this repo's plugin use the same top-level package names (they must: A1111 puts the
    extension root on sys.path), so the reference modules are imported, captured, and then *removed*
    from sys.modules again -- both can live in one process, under different Python objects.
    """
    global _REF_CACHE
    if _REF_CACHE is not None:
        return _REF_CACHE
    if not reference_available():
        raise FileNotFoundError(f"reference not mounted at {REFERENCE_ROOT}")
    install()
    mine = _evict(_PLUGIN_TOPLEVEL)
    old_path = list(sys.path)
    sys.path[:] = [REFERENCE_ROOT] + [q for q in sys.path if os.path.abspath(q or ".") != PLUGIN_ROOT]
    importlib.invalidate_caches()
    try:
        utils = importlib.import_module("tile_utils.utils")
        attn = importlib.import_module("tile_utils.attn")
        absd = importlib.import_module("tile_methods.abstractdiffusion")
        md = importlib.import_module("tile_methods.multidiffusion")
        mod = importlib.import_module("tile_methods.mixtureofdiffusers")
        try:
            demofusion = importlib.import_module("tile_methods.demofusion")
        except Exception as e:                                                   # pragma: no cover
            demofusion = None
            print(f"[stub_host] upstream tile_methods/demofusion.py not importable under the stub host: {e!r}")
        tilevae = importlib.import_module("scripts.tilevae")
        try:
            tilediffusion = importlib.import_module("scripts.tilediffusion")   # region-noise hijack (:486-529)
        except Exception as e:                                                   # pragma: no cover - depends on the stubs
            tilediffusion = None
            print(f"[stub_host] upstream scripts/tilediffusion.py not importable under the stub host: {e!r}")
    finally:
        sys.path[:] = old_path
        ref_mods = _evict(_PLUGIN_TOPLEVEL)
        sys.modules.update(mine)
        importlib.invalidate_caches()
    # bypass the `shared.sd_model.cond_stage_key` probe (abstractdiffusion.py:17-20)
    md.MultiDiffusion.is_edit_model = False
    mod.MixtureOfDiffusers.is_edit_model = False
    if demofusion is not None:
        demofusion.DemoFusion.is_edit_model = False
    _REF_CACHE = SimpleNamespace(utils=utils, attn=attn, abstractdiffusion=absd, multidiffusion=md, demofusion=demofusion,
                                 mixtureofdiffusers=mod, tilevae=tilevae, tilediffusion=tilediffusion, _modules=ref_mods)
    return _REF_CACHE


def load_plugin() -> SimpleNamespace:
    """Import this repo's plugin (the product) under the stub host."""
    install()
    if PLUGIN_ROOT not in sys.path:
        sys.path.insert(0, PLUGIN_ROOT)
    utils = importlib.import_module("tile_utils.utils")
    absd = importlib.import_module("tile_methods.abstractdiffusion")
    md = importlib.import_module("tile_methods.multidiffusion")
    mod = importlib.import_module("tile_methods.mixtureofdiffusers")
    tilevae = importlib.import_module("scripts.tilevae")
    tilediffusion = importlib.import_module("scripts.tilediffusion")
    demofusion = importlib.import_module("tile_methods.demofusion")
    tileglobal = importlib.import_module("scripts.tileglobal")
    engine = importlib.import_module("mdtile")
    return SimpleNamespace(utils=utils, abstractdiffusion=absd, multidiffusion=md, mixtureofdiffusers=mod, demofusion=demofusion,
                           tilevae=tilevae, tilediffusion=tilediffusion, tileglobal=tileglobal, engine=engine)


def make_processing(width: int, height: int, sampler_name: str = "Euler", **kw) -> SimpleNamespace:
    """A `p` that satisfies AbstractDiffusion.__init__ (abstractdiffusion.py:6-28)."""
    return SimpleNamespace(width=width, height=height, sampler_name=sampler_name, disable_extra_networks=True,
                           batch_size=1, steps=20, styles=[], all_prompts=[""], all_negative_prompts=[""], **kw)


def kdiff_sampler():
    return sys.modules["modules.sd_samplers_kdiffusion"].KDiffusionSampler()
