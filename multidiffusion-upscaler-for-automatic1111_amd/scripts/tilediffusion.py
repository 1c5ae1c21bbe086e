"""
Tiled Diffusion Script -- the A1111 plugin surface of upstream scripts/tilediffusion.py on top of the mdtile engine.

Kept verbatim: the Script name, `show`, the positional argument order of `process`
(= the order of the components `ui` returns), the hijack points (`sd_samplers.create_sampler`, and through the
delegates `sampler.model_wrap_cfg.inner_model.forward` / `shared.sd_model.apply_model`), the attribute names used to
stash originals (`create_sampler_original_md`, `apply_model_original_md`), the "Tiled Diffusion" infotext block and the
in-place restore of `p.width/height` in `postprocess`.

Per-region seeds (`processing.create_random_tensors` hijack, upstream :376-383, :486-529) run on the engine
(`mdtile_region_noise`); Noise Inversion (:431-450, renoise composite on `mdtile_noise_inverse_blend`) and the ControlNet /
StableSR tensor tiling (:344-361, `mdtile_gather_rects`) are armed exactly where upstream arms them.  Not carried over: the
region-config file dialog (UI only).
"""
from __future__ import annotations

from typing import Any, List

import torch

from modules import devices, processing, scripts, sd_samplers, shared
from modules.shared import opts, state  # noqa: F401
from modules.processing import opt_f

from tile_methods.abstractdiffusion import AbstractDiffusion
from tile_methods.mixtureofdiffusers import MixtureOfDiffusers
from tile_methods.multidiffusion import MultiDiffusion
from tile_utils.utils import (BlendMode, DEFAULT_BBOX_SETTINGS, Method, NUM_BBOX_PARAMS, build_bbox_settings, splitable)

CFG_PATH_NOTE = "region_configs"
BBOX_MAX_NUM = min(getattr(shared.cmd_opts, "md_max_regions", 8), 16)


class Script(scripts.Script):

    def __init__(self):
        self.controlnet_script = None
        self.stablesr_script = None
        self.delegate: AbstractDiffusion = None
        self.noise_inverse_cache = None

    def title(self):
        return "Tiled Diffusion"

    def show(self, is_img2img):
        return scripts.AlwaysVisible

    def ui(self, is_img2img):
        import gradio as gr
        tab = "t2i" if not is_img2img else "i2i"
        uid = lambda name: f"MD-{tab}-{name}"  # noqa: E731
        with gr.Accordion("Tiled Diffusion", open=False, elem_id=f"MD-{tab}"):
            with gr.Row(variant="compact"):
                enabled = gr.Checkbox(label="Enable Tiled Diffusion", value=False, elem_id=uid("enable"))
                overwrite_size = gr.Checkbox(label="Overwrite image size", value=False, visible=not is_img2img, elem_id=uid("overwrite-image-size"))
                keep_input_size = gr.Checkbox(label="Keep input image size", value=True, visible=is_img2img, elem_id=uid("keep-input-size"))
            with gr.Row(variant="compact", visible=False):
                image_width = gr.Slider(minimum=256, maximum=16384, step=16, label="Image width", value=1024, elem_id=f"MD-overwrite-width-{tab}")
                image_height = gr.Slider(minimum=256, maximum=16384, step=16, label="Image height", value=1024, elem_id=f"MD-overwrite-height-{tab}")
            with gr.Row(variant="compact"):
                method = gr.Dropdown(label="Method", choices=[e.value for e in Method], value=Method.MULTI_DIFF.value, elem_id=uid("method"))
                control_tensor_cpu = gr.Checkbox(label="Move ControlNet tensor to CPU (if applicable)", value=False, elem_id=uid("control-tensor-cpu"))
            with gr.Row(variant="compact"):
                tile_width = gr.Slider(minimum=16, maximum=256, step=16, label="Latent tile width", value=96, elem_id=uid("latent-tile-width"))
                tile_height = gr.Slider(minimum=16, maximum=256, step=16, label="Latent tile height", value=96, elem_id=uid("latent-tile-height"))
            with gr.Row(variant="compact"):
                overlap = gr.Slider(minimum=0, maximum=256, step=4, label="Latent tile overlap", value=48 if not is_img2img else 8, elem_id=uid("latent-tile-overlap"))
                batch_size = gr.Slider(minimum=1, maximum=8, step=1, label="Latent tile batch size", value=4, elem_id=uid("latent-tile-batch-size"))
            with gr.Row(variant="compact", visible=is_img2img):
                upscaler_name = gr.Dropdown(label="Upscaler", choices=[x.name for x in shared.sd_upscalers], value="None", elem_id=uid("upscaler-index"))
                scale_factor = gr.Slider(minimum=1.0, maximum=8.0, step=0.05, label="Scale Factor", value=2.0, elem_id=uid("upscaler-factor"))
            with gr.Accordion("Noise Inversion", open=True, visible=is_img2img):
                noise_inverse = gr.Checkbox(label="Enable Noise Inversion", value=False, elem_id=uid("noise-inverse"))
                noise_inverse_steps = gr.Slider(minimum=1, maximum=200, step=1, label="Inversion steps", value=10, elem_id=uid("noise-inverse-steps"))
                noise_inverse_retouch = gr.Slider(minimum=1, maximum=100, step=0.1, label="Retouch", value=1, elem_id=uid("noise-inverse-retouch"))
                noise_inverse_renoise_strength = gr.Slider(minimum=0, maximum=2, step=0.01, label="Renoise strength", value=1, elem_id=uid("noise-inverse-renoise-strength"))
                noise_inverse_renoise_kernel = gr.Slider(minimum=2, maximum=512, step=1, label="Renoise kernel size", value=64, elem_id=uid("noise-inverse-renoise-kernel"))
            with gr.Group(elem_id=f"MD-bbox-control-{tab}"):
                with gr.Accordion("Region Prompt Control", open=False):
                    with gr.Row(variant="compact"):
                        enable_bbox_control = gr.Checkbox(label="Enable Control", value=False, elem_id=uid("enable-bbox-control"))
                        draw_background = gr.Checkbox(label="Draw full canvas background", value=False, elem_id=uid("draw-background"))
                        causal_layers = gr.Checkbox(label="Causalize layers", value=False, visible=False, elem_id=uid("causal-layers"))
                    bbox_controls: List[Any] = []
                    for i in range(BBOX_MAX_NUM):
                        with gr.Accordion(f"Region {i + 1}", open=False, elem_id=f"MD-accordion-{tab}-{i}"):
                            e = gr.Checkbox(label=f"Enable Region {i + 1}", value=False, elem_id=f"MD-bbox-{tab}-{i}-enable")
                            x = gr.Slider(label="x", value=0.4, minimum=0.0, maximum=1.0, step=0.0001, elem_id=f"MD-{tab}-{i}-x")
                            y = gr.Slider(label="y", value=0.4, minimum=0.0, maximum=1.0, step=0.0001, elem_id=f"MD-{tab}-{i}-y")
                            w = gr.Slider(label="w", value=0.2, minimum=0.0, maximum=1.0, step=0.0001, elem_id=f"MD-{tab}-{i}-w")
                            h = gr.Slider(label="h", value=0.2, minimum=0.0, maximum=1.0, step=0.0001, elem_id=f"MD-{tab}-{i}-h")
                            prompt = gr.Text(show_label=False, placeholder="Prompt, will append to your main prompt", max_lines=2, elem_id=f"MD-{tab}-{i}-prompt")
                            neg_prompt = gr.Text(show_label=False, placeholder="Negative Prompt, will also be appended", max_lines=1, elem_id=f"MD-{tab}-{i}-neg-prompt")
                            blend_mode = gr.Dropdown(label="Type", choices=[e_.value for e_ in BlendMode], value=BlendMode.BACKGROUND.value, elem_id=f"MD-{tab}-{i}-blend-mode")
                            feather_ratio = gr.Slider(label="Feather", value=0.2, minimum=0, maximum=1, step=0.05, elem_id=f"MD-{tab}-{i}-feather")
                            seed = gr.Number(label="Seed", value=-1, precision=0, elem_id=f"MD-{tab}-{i}-seed")
                        control = [e, x, y, w, h, prompt, neg_prompt, blend_mode, feather_ratio, seed]
                        assert len(control) == NUM_BBOX_PARAMS
                        bbox_controls.extend(control)
        return [
            enabled, method,
            overwrite_size, keep_input_size, image_width, image_height,
            tile_width, tile_height, overlap, batch_size,
            upscaler_name, scale_factor,
            noise_inverse, noise_inverse_steps, noise_inverse_retouch, noise_inverse_renoise_strength, noise_inverse_renoise_kernel,
            control_tensor_cpu,
            enable_bbox_control, draw_background, causal_layers,
            *bbox_controls,
        ]

    def process(self, p,
                enabled: bool, method: str,
                overwrite_size: bool, keep_input_size: bool, image_width: int, image_height: int,
                tile_width: int, tile_height: int, overlap: int, tile_batch_size: int,
                upscaler_name: str, scale_factor: float,
                noise_inverse: bool, noise_inverse_steps: int, noise_inverse_retouch: float,
                noise_inverse_renoise_strength: float, noise_inverse_renoise_kernel: int,
                control_tensor_cpu: bool,
                enable_bbox_control: bool, draw_background: bool, causal_layers: bool,
                *bbox_control_states: List[Any]):
        self.reset()   # undo leftovers of a job that died half-way
        if not enabled:
            return

        # canvas size bookkeeping: stash the originals unconditionally, postprocess restores them (upstream :272-306, :396-402)
        if hasattr(p, "init_images"):
            p.init_images_original_md = [img.copy() for img in p.init_images]
        p.width_original_md, p.height_original_md = p.width, p.height

        is_img2img = hasattr(p, "init_images") and len(getattr(p, "init_images", []) or []) > 0
        upscaled = False
        if is_img2img:
            names = [x.name for x in shared.sd_upscalers]
            upscaler = shared.sd_upscalers[names.index(upscaler_name)] if upscaler_name in names else None
            image = p.init_images[0]
            try:
                from modules import images
                image = images.flatten(image, opts.img2img_background_color)
            except Exception:   # bare test host: nothing to flatten
                pass
            if upscaler is not None and upscaler.name != "None":
                print(f"[Tiled Diffusion] upscaling image with {upscaler.name}...")
                image = upscaler.scaler.upscale(image, scale_factor, upscaler.data_path)
                p.extra_generation_params["Tiled Diffusion upscaler"] = upscaler.name
                p.extra_generation_params["Tiled Diffusion scale factor"] = scale_factor
                for i in range(len(p.init_images)):       # folder-based batches hold several entries: all become the upscaled image
                    p.init_images[i] = image
                upscaled = True
            if keep_input_size:
                p.width, p.height = image.width, image.height
            elif upscaled:
                p.width, p.height = int(scale_factor * p.width_original_md), int(scale_factor * p.height_original_md)
        elif overwrite_size:
            p.width, p.height = image_width, image_height

        bbox_settings = build_bbox_settings(bbox_control_states) if enable_bbox_control else {}
        if not (splitable(p.width, p.height, tile_width, tile_height, overlap) or enable_bbox_control or (is_img2img and noise_inverse)):
            print("[Tiled Diffusion] ignored: the image fits one tile and there is nothing else to do.")
            return

        info = {"Method": method, "Tile tile width": tile_width, "Tile tile height": tile_height,
                "Tile Overlap": overlap, "Tile batch size": tile_batch_size}
        if is_img2img:
            if upscaled:
                info["Upscaler"], info["Upscale factor"] = upscaler.name, scale_factor
            if keep_input_size:
                info["Keep input size"] = keep_input_size
            if noise_inverse:
                info.update({"NoiseInv": noise_inverse, "NoiseInv Steps": noise_inverse_steps, "NoiseInv Retouch": noise_inverse_retouch,
                             "NoiseInv Renoise strength": noise_inverse_renoise_strength, "NoiseInv Kernel size": noise_inverse_renoise_kernel})
        if bbox_settings:
            info["Region control"] = {f"Region {i + 1}": s._asdict() for i, s in bbox_settings.items()}
        if not hasattr(p, "extra_generation_params") or p.extra_generation_params is None:
            p.extra_generation_params = {}
        p.extra_generation_params["Tiled Diffusion"] = info

        # other extensions whose tensors have to follow the tiles (upstream :344-361)
        self.controlnet_script = self.stablesr_script = None
        runner = getattr(p, "scripts", None)
        if runner is not None:
            try:
                import scripts.cldm  # noqa: F401   (only tells whether sd-webui-controlnet is installed)
                for sc in list(getattr(runner, "scripts", [])) + list(getattr(runner, "alwayson_scripts", [])):
                    if hasattr(sc, "latest_network") and sc.title().lower() == "controlnet":
                        self.controlnet_script = sc
                        print("[Tiled Diffusion] ControlNet found, support is enabled.")
                        break
            except ImportError:
                pass
            for sc in getattr(runner, "scripts", []):
                if hasattr(sc, "stablesr_model") and sc.title().lower() == "stablesr" and sc.stablesr_model is not None:
                    self.stablesr_script = sc
                    print("[Tiled Diffusion] StableSR found, support is enabled.")
                    break

        Script.create_sampler_original_md = sd_samplers.create_sampler
        sd_samplers.create_sampler = lambda name, model: self.create_sampler_hijack(
            name, model, p, Method(method), tile_width, tile_height, overlap, tile_batch_size,
            noise_inverse, noise_inverse_steps, noise_inverse_retouch, noise_inverse_renoise_strength, noise_inverse_renoise_kernel,
            control_tensor_cpu, enable_bbox_control, draw_background, causal_layers, bbox_settings)

        if enable_bbox_control:
            # every region gets its own seeded initial noise (upstream :376-383)
            region_info = info.setdefault("Region control", {})
            Script.create_random_tensors_original_md = processing.create_random_tensors
            processing.create_random_tensors = lambda *args, **kwargs: self.create_random_tensors_hijack(
                bbox_settings, region_info, *args, **kwargs)

    def postprocess_batch(self, p, enabled, *args, **kwargs):
        if enabled and self.delegate is not None:
            self.delegate.reset_controlnet_tensors()

    def postprocess(self, p, processed, enabled, *args):
        if not enabled:
            return
        self.reset()
        if hasattr(p, "init_images") and hasattr(p, "init_images_original_md"):
            p.init_images.clear()       # keep the list OBJECT: XYZ-plot works on shallow copies of p
            p.init_images.extend(p.init_images_original_md)
            del p.init_images_original_md
        if hasattr(p, "width_original_md"):
            p.width, p.height = p.width_original_md, p.height_original_md
            del p.width_original_md, p.height_original_md
        if hasattr(p, "noise_inverse_latent"):
            del p.noise_inverse_latent

    # ---- hijack ---------------------------------------------------------------------------------------------------
    def create_sampler_hijack(self, name: str, model, p, method: Method, tile_width: int, tile_height: int, overlap: int,
                              tile_batch_size: int, noise_inverse: bool, noise_inverse_steps: int, noise_inverse_retouch: float,
                              noise_inverse_renoise_strength: float, noise_inverse_renoise_kernel: int, control_tensor_cpu: bool,
                              enable_bbox_control: bool, draw_background: bool, causal_layers: bool, bbox_settings):
        if self.delegate is not None and self.delegate.sampler_name == name:
            # second call within one job (e.g. hires fix, ControlNet batches): keep the delegate, refresh what follows the tiles
            if self.controlnet_script:
                self.delegate.prepare_controlnet_tensors(refresh=True)
            if isinstance(self.delegate, MixtureOfDiffusers):
                self.delegate.hook()
            return self.delegate.sampler_raw
        self.reset(keep_sampler_hijack=True)

        flag_noise_inverse = hasattr(p, "init_images") and len(p.init_images) > 0 and noise_inverse
        if flag_noise_inverse:
            print('[Tiled Diffusion] Noise Inversion only supports the "Euler" sampler: switching to it.')
            name = "Euler"
            p.sampler_name = "Euler"
        sampler = Script.create_sampler_original_md(name, model)
        cls = MultiDiffusion if method == Method.MULTI_DIFF else MixtureOfDiffusers
        delegate = cls(p, sampler)
        if flag_noise_inverse:
            delegate.init_noise_inverse(noise_inverse_steps, noise_inverse_retouch, self.noise_inverse_get_cache,
                                        lambda x0, xt, prompts: self.noise_inverse_set_cache(p, x0, xt, prompts, noise_inverse_steps, noise_inverse_retouch),
                                        noise_inverse_renoise_strength, noise_inverse_renoise_kernel)
        if not enable_bbox_control or draw_background:
            delegate.init_grid_bbox(tile_width, tile_height, overlap, tile_batch_size)
        if enable_bbox_control and bbox_settings:
            delegate.init_custom_bbox(bbox_settings, draw_background, causal_layers)
        if self.controlnet_script:
            delegate.init_controlnet(self.controlnet_script, control_tensor_cpu)
        if self.stablesr_script:
            delegate.init_stablesr(self.stablesr_script)
        delegate.init_done()
        delegate.hook()
        self.delegate = delegate

        exts = [n for n, on in (("NoiseInv", flag_noise_inverse), ("RegionCtrl", enable_bbox_control), ("ControlNet", bool(self.controlnet_script)),
                                ("StableSR", bool(self.stablesr_script))) if on]
        print(f"[Tiled Diffusion] {method.value} hooked into {name!r} sampler; tile {tile_width}x{tile_height}, "
              f"overlap {overlap}, batch {tile_batch_size}, {delegate.num_tiles or 0} tiles in "
              f"{delegate.num_batches or 0} batches" + (f", {len(delegate.custom_bboxes)} regions" if delegate.custom_bboxes else "")
              + (f"; ext: {', '.join(exts)}" if exts else ""))
        return delegate.sampler_raw

    def noise_inverse_set_cache(self, p, x0, xt, prompts, steps: int, retouch: float):
        from tile_utils.utils import NoiseInverseCache
        self.noise_inverse_cache = NoiseInverseCache(p.sd_model.sd_model_hash, x0, xt, steps, retouch, prompts)

    def noise_inverse_get_cache(self):
        return self.noise_inverse_cache

    def create_random_tensors_hijack(self, bbox_settings, region_info, shape, seeds, subseeds=None, subseed_strength=0.0,
                                     seed_resize_from_h=0, seed_resize_from_w=0, p=None):
        """Upstream :486-529: the job's noise, with every region's rectangle replaced by that region's own noise
        (`torch.manual_seed(seed)` + CPU `randn`, exactly as upstream draws it); overlapping regions of one kind are
        averaged, foreground goes on top of background.  The sum / count / average / paste runs in ONE engine launch."""
        import math
        import mdtile
        from modules.processing import get_fixed_seed
        noise = Script.create_random_tensors_original_md(shape, seeds, subseeds, subseed_strength, seed_resize_from_h,
                                                         seed_resize_from_w, p)
        height, width = shape[1], shape[2]
        regions = []
        for i, v in bbox_settings.items():
            seed = get_fixed_seed(v.seed)
            x, y = max(0, int(v.x * width)), max(0, int(v.y * height))
            w, h = min(width - x, math.ceil(v.w * width)), min(height - y, math.ceil(v.h * height))
            torch.manual_seed(seed)
            rand = torch.randn((1, noise.shape[1], h, w), device=devices.cpu)
            mode = BlendMode(v.blend_mode)
            if mode not in (BlendMode.BACKGROUND, BlendMode.FOREGROUND):
                raise NotImplementedError
            regions.append((x, y, w, h, mdtile.REGION_BG if mode == BlendMode.BACKGROUND else mdtile.REGION_FG,
                            rand.to(device=noise.device, dtype=torch.float32).contiguous()))
            key = "Region " + str(i + 1)
            if key in region_info:
                region_info[key]["seed"] = seed
        work = noise.to(torch.float32).contiguous()
        mdtile.region_noise(work, regions)
        return work.to(noise.dtype)

    def reset(self, keep_sampler_hijack: bool = False):
        if not keep_sampler_hijack and hasattr(Script, "create_sampler_original_md"):
            sd_samplers.create_sampler = Script.create_sampler_original_md
            del Script.create_sampler_original_md
        if not keep_sampler_hijack and hasattr(Script, "create_random_tensors_original_md"):
            processing.create_random_tensors = Script.create_random_tensors_original_md
            del Script.create_random_tensors_original_md
        MultiDiffusion.unhook()
        MixtureOfDiffusers.unhook()
        self.delegate = None
