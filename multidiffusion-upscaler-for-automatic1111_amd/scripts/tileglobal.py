"""
DemoFusion Script -- the A1111 plugin surface of upstream scripts/tileglobal.py on top of the mdtile engine: same Script title
('demofusion'), the positional argument order of `process` (= the components `ui` returns), the hijack points
(`sd_samplers.create_sampler`, `p.sample`, `processing.create_infotext`, `Sampler.callback_state`) and the progressive
upscaling loop of `sample_hijack` (phase k: bicubic x k of the previous latent, re-noise, tiled + dilated denoising by the
DemoFusion delegate, re-standardise to the first phase's statistics).  The per-evaluation arithmetic lives in
tile_methods/demofusion.py / csrc/demofusion.hip; this file is host control flow.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from modules import devices, processing, scripts, sd_samplers, sd_samplers_common, shared
from modules.shared import opts, state  # noqa: F401
from modules.processing import opt_f

from tile_methods.abstractdiffusion import AbstractDiffusion
from tile_methods.demofusion import DemoFusion
from tile_utils.utils import Method_2, NoiseInverseCache


def create_infotext_hijack(p, all_prompts, all_seeds, all_subseeds, comments=None, iteration=0, position_in_batch=0,
                           use_main_prompt=False, index=-1, all_negative_prompts=None):
    """Every phase returns an image of its own size: patch the 'Size' field of the infotext per image (upstream :27-39)."""
    idx = None if index == -1 else index
    text = processing.create_infotext_ori(p, all_prompts, all_seeds, all_subseeds, comments, iteration, position_in_batch,
                                          use_main_prompt, idx, all_negative_prompts)
    a = text.find("Size")
    if a != -1:
        b = text.find(",", a)
        if b != -1:
            return text[:a] + f"Size:{p.width_list[index]}x{p.height_list[index]}" + text[b:]
    return text


class Script(scripts.Script):

    def __init__(self):
        self.controlnet_script = None
        self.stablesr_script = None
        self.delegate: AbstractDiffusion = None
        self.noise_inverse_cache = None
        self.flag_noise_inverse = False

    def title(self):
        return "demofusion"

    def show(self, is_img2img):
        return scripts.AlwaysVisible

    def ui(self, is_img2img):
        import gradio as gr
        tab = "demofusion-t2i" if not is_img2img else "demofusion-i2i"
        uid = lambda name: f"MD-{tab}-{name}"  # noqa: E731
        with gr.Accordion("DemoFusion", open=False, elem_id=f"MD-{tab}"):
            with gr.Row(variant="compact"):
                enabled = gr.Checkbox(label="Enable DemoFusion(Dont open with tilediffusion)", value=False, elem_id=uid("enabled"))
                random_jitter = gr.Checkbox(label="Random Jitter", value=True, elem_id=uid("random-jitter"))
                keep_input_size = gr.Checkbox(label="Keep input-image size", value=False, visible=is_img2img, elem_id=uid("keep-input-size"))
                mixture_mode = gr.Checkbox(label="Mixture mode", value=False, elem_id=uid("mixture-mode"))
                gaussian_filter = gr.Checkbox(label="Gaussian Filter", value=True, visible=False, elem_id=uid("gaussian"))
            with gr.Row(variant="compact"):
                method = gr.Dropdown(label="Method", choices=[Method_2.DEMO_FU.value], value=Method_2.DEMO_FU.value, visible=False, elem_id=uid("method"))
                control_tensor_cpu = gr.Checkbox(label="Move ControlNet tensor to CPU (if applicable)", value=False, elem_id=uid("control-tensor-cpu"))
            with gr.Row(variant="compact"):
                window_size = gr.Slider(minimum=16, maximum=256, step=16, label="Latent window size", value=128, elem_id=uid("latent-window-size"))
            with gr.Row(variant="compact"):
                overlap = gr.Slider(minimum=0, maximum=256, step=4, label="Latent window overlap", value=64, elem_id=uid("latent-tile-overlap"))
                batch_size = gr.Slider(minimum=1, maximum=8, step=1, label="Latent window batch size", value=4, elem_id=uid("latent-tile-batch-size"))
                batch_size_g = gr.Slider(minimum=1, maximum=8, step=1, label="Global window batch size", value=4, elem_id=uid("Global-tile-batch-size"))
            with gr.Row(variant="compact"):
                c1 = gr.Slider(minimum=0, maximum=5, step=0.01, label="Cosine Scale 1", value=3, elem_id=f"C1-{tab}")
                c2 = gr.Slider(minimum=0, maximum=5, step=0.01, label="Cosine Scale 2", value=1, elem_id=f"C2-{tab}")
                c3 = gr.Slider(minimum=0, maximum=5, step=0.01, label="Cosine Scale 3", value=1, elem_id=f"C3-{tab}")
                sigma = gr.Slider(minimum=0, maximum=2, step=0.01, label="Sigma", value=0.6, elem_id=f"Sigma-{tab}")
            strength = gr.Slider(minimum=0, maximum=1, step=0.01, value=0.85, label="Denoising Strength for Substage", visible=not is_img2img, elem_id=f"strength-{tab}")
            scale_factor = gr.Slider(minimum=1.0, maximum=8.0, step=1, label="Scale Factor", value=2.0, elem_id=uid("upscaler-factor"))
            with gr.Accordion("Noise Inversion", open=True, visible=is_img2img):
                noise_inverse = gr.Checkbox(label="Enable Noise Inversion", value=False, elem_id=uid("noise-inverse"))
                noise_inverse_steps = gr.Slider(minimum=1, maximum=200, step=1, label="Inversion steps", value=10, elem_id=uid("noise-inverse-steps"))
                noise_inverse_retouch = gr.Slider(minimum=1, maximum=100, step=0.1, label="Retouch", value=1, elem_id=uid("noise-inverse-retouch"))
                noise_inverse_renoise_strength = gr.Slider(minimum=0, maximum=2, step=0.01, label="Renoise strength", value=1, elem_id=uid("noise-inverse-renoise-strength"))
                noise_inverse_renoise_kernel = gr.Slider(minimum=2, maximum=512, step=1, label="Renoise kernel size", value=64, elem_id=uid("noise-inverse-renoise-kernel"))
        return [
            enabled, method,
            keep_input_size,
            window_size, overlap, batch_size,
            scale_factor,
            noise_inverse, noise_inverse_steps, noise_inverse_retouch, noise_inverse_renoise_strength, noise_inverse_renoise_kernel,
            control_tensor_cpu,
            random_jitter,
            c1, c2, c3, gaussian_filter, strength, sigma, batch_size_g, mixture_mode,
        ]

    def process(self, p, enabled: bool, method: str, keep_input_size: bool, window_size: int, overlap: int, tile_batch_size: int,
                scale_factor: float, noise_inverse: bool, noise_inverse_steps: int, noise_inverse_retouch: float,
                noise_inverse_renoise_strength: float, noise_inverse_renoise_kernel: int, control_tensor_cpu: bool, random_jitter: bool,
                c1, c2, c3, gaussian_filter, strength, sigma, batch_size_g, mixture_mode):
        self.reset()
        p.mixture = mixture_mode
        if not mixture_mode:
            sigma = sigma / 2
        if not enabled:
            return
        if hasattr(p, "init_images"):
            p.init_images_original_md = [img.copy() for img in p.init_images]
        p.width_original_md, p.height_original_md = p.width, p.height
        p.current_scale_num = 1
        p.gaussian_filter = gaussian_filter
        p.scale_factor = int(scale_factor)
        is_img2img = hasattr(p, "init_images") and len(p.init_images) > 0
        if is_img2img and keep_input_size:
            image = p.init_images[0]
            try:
                from modules import images
                image = images.flatten(image, opts.img2img_background_color)
            except Exception:
                pass
            p.width, p.height = image.width, image.height
            p.width_original_md, p.height_original_md = p.width, p.height

        if not hasattr(p, "extra_generation_params") or p.extra_generation_params is None:
            p.extra_generation_params = {}
        info = {"Method": method, "Window Size": window_size, "Tile Overlap": overlap, "Tile batch size": tile_batch_size,
                "Global batch size": batch_size_g}
        if is_img2img:
            info["Upscale factor"] = scale_factor
            if keep_input_size:
                info["Keep input size"] = keep_input_size
            if noise_inverse:
                info.update({"NoiseInv": noise_inverse, "NoiseInv Steps": noise_inverse_steps, "NoiseInv Retouch": noise_inverse_retouch,
                             "NoiseInv Renoise strength": noise_inverse_renoise_strength, "NoiseInv Kernel size": noise_inverse_renoise_kernel})
        p.extra_generation_params["Tiled Diffusion"] = info

        self.controlnet_script = self.stablesr_script = None
        runner = getattr(p, "scripts", None)
        if runner is not None:
            try:
                import scripts.cldm  # noqa: F401
                for sc in list(getattr(runner, "scripts", [])) + list(getattr(runner, "alwayson_scripts", [])):
                    if hasattr(sc, "latest_network") and sc.title().lower() == "controlnet":
                        self.controlnet_script = sc
                        break
            except ImportError:
                pass
            for sc in getattr(runner, "scripts", []):
                if hasattr(sc, "stablesr_model") and sc.title().lower() == "stablesr" and sc.stablesr_model is not None:
                    self.stablesr_script = sc
                    break

        Script.create_sampler_original_md = sd_samplers.create_sampler
        sd_samplers.create_sampler = lambda name, model: self.create_sampler_hijack(
            name, model, p, Method_2(method), control_tensor_cpu, window_size, noise_inverse, noise_inverse_steps, noise_inverse_retouch,
            noise_inverse_renoise_strength, noise_inverse_renoise_kernel, overlap, tile_batch_size, random_jitter, batch_size_g)
        p.sample = lambda conditioning, unconditional_conditioning, seeds, subseeds, subseed_strength, prompts: self.sample_hijack(
            conditioning, unconditional_conditioning, seeds, subseeds, subseed_strength, prompts, p, is_img2img, window_size, overlap,
            tile_batch_size, random_jitter, c1, c2, c3, strength, sigma, batch_size_g)
        processing.create_infotext_ori = processing.create_infotext
        p.width_list, p.height_list = [p.height], [p.height]
        processing.create_infotext = create_infotext_hijack

    def postprocess_batch(self, p, enabled, *args, **kwargs):
        if enabled and self.delegate is not None:
            self.delegate.reset_controlnet_tensors()

    def postprocess_batch_list(self, p, pp, enabled, *args, **kwargs):
        """The phases' latents were stacked zero-padded to the final size: crop every decoded image to its own phase (upstream :236-250)."""
        if not enabled:
            return
        sf = p.scale_factor
        for idx, image in enumerate(pp.images):
            k = idx // p.batch_size + 1
            pp.images[idx] = image[:, :image.shape[1] // sf * k, :image.shape[2] // sf * k]
        p.seeds = [s for _ in range(sf) for s in p.seeds]
        p.prompts = [s for _ in range(sf) for s in p.prompts]
        p.all_negative_prompts = [s for _ in range(sf) for s in p.all_negative_prompts]
        p.negative_prompts = [s for _ in range(sf) for s in p.negative_prompts]
        if getattr(p, "color_corrections", None) is not None:
            p.color_corrections = [s for _ in range(sf) for s in p.color_corrections]
        p.width_list = [p.width * (k + 1) for k in range(sf) for _ in range(p.batch_size)]
        p.height_list = [p.height * (k + 1) for k in range(sf) for _ in range(p.batch_size)]

    def postprocess(self, p, processed, enabled, *args):
        if not enabled:
            return
        self.reset()
        if hasattr(p, "init_images") and hasattr(p, "init_images_original_md"):
            p.init_images.clear()
            p.init_images.extend(p.init_images_original_md)
            del p.init_images_original_md
        p.width, p.height = p.width_original_md, p.height_original_md
        del p.width_original_md, p.height_original_md
        if hasattr(p, "noise_inverse_latent"):
            del p.noise_inverse_latent

    # ---- hijacks -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_hijack(self, conditioning, unconditional_conditioning, seeds, subseeds, subseed_strength, prompts, p, image_ori,
                      window_size, overlap, tile_batch_size, random_jitter, c1, c2, c3, strength, sigma, batch_size_g):
        """Phase 1: an ordinary sample at the base size (txt2img) or the encoded init image (img2img).  Phase k = 2 .. scale_factor:
        bicubic upscale of the latent, fresh noise, img2img denoising through the DemoFusion delegate, statistics pulled back to
        phase 1's.  Returns the phases stacked along the batch axis, zero-padded to the last phase's size (upstream :272-372)."""
        from modules import rng
        if not image_ori:
            p.current_step = 0
            p.denoising_strength = strength
            p.sampler = Script.create_sampler_original_md(p.sampler_name, p.sd_model)
            x = p.rng.next()
            print("### Phase 1 Denoising ###")
            latents = p.sampler.sample(p, x, conditioning, unconditional_conditioning, image_conditioning=p.txt2img_image_conditioning(x))
            res = F.pad(latents, (0, latents.shape[3] * (p.scale_factor - 1), 0, latents.shape[2] * (p.scale_factor - 1)))
            del x
            p.sampler = sd_samplers.create_sampler(p.sampler_name, p.sd_model)
            starting_scale = 2
        else:
            print("### Encoding Real Image ###")
            latents = p.init_latent
            res = None
            starting_scale = 1
        anchor_mean, anchor_std = latents.mean(), latents.std()
        devices.torch_gc()

        p.cosine_scale_1, p.cosine_scale_2, p.cosine_scale_3 = c1, c2, c3
        self.delegate.sig = sigma
        p.latents = latents
        for k in range(starting_scale, p.scale_factor + 1):
            p.current_scale_num = k
            print(f"### Phase {k} Denoising ###")
            p.current_height, p.current_width = p.height_original_md * k, p.width_original_md * k
            p.latents = F.interpolate(p.latents, size=(int(p.current_height / opt_f), int(p.current_width / opt_f)), mode="bicubic")
            p.rng = rng.ImageRNG(p.latents.shape[1:], p.seeds, subseeds=p.subseeds, subseed_strength=p.subseed_strength,
                                 seed_resize_from_h=p.seed_resize_from_h, seed_resize_from_w=p.seed_resize_from_w)
            self.delegate.w, self.delegate.h = int(p.current_width / opt_f), int(p.current_height / opt_f)
            self.delegate.get_views(overlap, tile_batch_size, batch_size_g)
            d = self.delegate
            print(f"Tile size: {d.window_size}, Tile count: {d.num_tiles}, Batch size: {d.tile_bs}, Tile batches: {len(d.batched_bboxes)}, "
                  f"Global batch size: {d.global_tile_bs}, Global batches: {len(d.global_batched_bboxes)}")
            noise = p.rng.next()
            if hasattr(p, "initial_noise_multiplier"):
                if p.initial_noise_multiplier != 1.0:
                    p.extra_generation_params["Noise multiplier"] = p.initial_noise_multiplier
                    noise *= p.initial_noise_multiplier
            else:
                p.image_conditioning = p.txt2img_image_conditioning(noise)
            p.noise = noise
            p.x = p.latents.clone()
            p.current_step = 0
            p.latents = p.sampler.sample_img2img(p, p.latents, noise, conditioning, unconditional_conditioning, image_conditioning=p.image_conditioning)
            if self.flag_noise_inverse:      # the inversion only seeds the first upscaling phase
                self.delegate.sampler_raw.sample_img2img = self.delegate.sample_img2img_original
                self.flag_noise_inverse = False
            p.latents = (p.latents - p.latents.mean()) / p.latents.std() * anchor_std + anchor_mean
            padded = F.pad(p.latents, (0, p.latents.shape[3] // k * (p.scale_factor - k), 0, p.latents.shape[2] // k * (p.scale_factor - k)))
            res = padded if res is None else torch.cat((res, padded), dim=0)
        return res

    @staticmethod
    def callback_hijack(self_sampler, d, p):
        p.current_step = d["i"]
        if self_sampler.stop_at is not None and p.current_step > self_sampler.stop_at:
            raise sd_samplers_common.InterruptedException
        state.sampling_step = p.current_step
        shared.total_tqdm.update()
        p.current_step += 1

    def create_sampler_hijack(self, name, model, p, method, control_tensor_cpu, window_size, noise_inverse, noise_inverse_steps,
                              noise_inverse_retouch, noise_inverse_renoise_strength, noise_inverse_renoise_kernel, overlap, tile_batch_size,
                              random_jitter, batch_size_g):
        if self.delegate is not None:
            if self.delegate.sampler_name == name:
                if self.controlnet_script:
                    self.delegate.prepare_controlnet_tensors(refresh=True)
                return self.delegate.sampler_raw
            self.reset()
        sd_samplers_common.Sampler.callback_ori = sd_samplers_common.Sampler.callback_state
        sd_samplers_common.Sampler.callback_state = lambda self_sampler, d: Script.callback_hijack(self_sampler, d, p)

        self.flag_noise_inverse = hasattr(p, "init_images") and len(p.init_images) > 0 and noise_inverse
        if self.flag_noise_inverse:
            print('[DemoFusion] Noise Inversion only supports the "Euler" sampler: switching to it.')
            name = "Euler"
            p.sampler_name = "Euler"
        sampler = Script.create_sampler_original_md(name, model)
        if method != Method_2.DEMO_FU:
            raise NotImplementedError(f"Method {method} not implemented.")
        delegate = DemoFusion(p, sampler)
        delegate.window_size = min(min(window_size, p.width // 8), p.height // 8)
        p.random_jitter = random_jitter
        if self.flag_noise_inverse:
            delegate.init_noise_inverse(noise_inverse_steps, noise_inverse_retouch, self.noise_inverse_get_cache,
                                        lambda x0, xt, prompts: self.noise_inverse_set_cache(p, x0, xt, prompts, noise_inverse_steps, noise_inverse_retouch),
                                        noise_inverse_renoise_strength, noise_inverse_renoise_kernel)
        if self.controlnet_script:
            delegate.init_controlnet(self.controlnet_script, control_tensor_cpu)
        if self.stablesr_script:
            delegate.init_stablesr(self.stablesr_script)
        delegate.hook()
        self.delegate = delegate
        print(f"[DemoFusion] {method.value} hooked into {name!r} sampler, window {delegate.window_size}")
        return delegate.sampler_raw

    def noise_inverse_set_cache(self, p, x0, xt, prompts, steps: int, retouch: float):
        self.noise_inverse_cache = NoiseInverseCache(p.sd_model.sd_model_hash, x0, xt, steps, retouch, prompts)

    def noise_inverse_get_cache(self):
        return self.noise_inverse_cache

    def reset(self):
        if hasattr(Script, "create_sampler_original_md"):
            sd_samplers.create_sampler = Script.create_sampler_original_md
            del Script.create_sampler_original_md
        if hasattr(processing, "create_infotext_ori"):
            processing.create_infotext = processing.create_infotext_ori
            del processing.create_infotext_ori
        if hasattr(sd_samplers_common.Sampler, "callback_ori"):
            sd_samplers_common.Sampler.callback_state = sd_samplers_common.Sampler.callback_ori
            del sd_samplers_common.Sampler.callback_ori
        DemoFusion.unhook()
        self.delegate = None
