"""The A1111 plugin surface the host binds to (SURVEY.md section 8b outer boundary) -- no GPU needed."""
import inspect

import pytest

from hostsim import stub_host as sh


def test_script_titles_and_visibility(plugin):
    td, tv = plugin.tilediffusion.Script(), plugin.tilevae.Script()
    assert td.title() == "Tiled Diffusion" and tv.title() == "Tiled VAE"
    import modules.scripts as scripts
    assert td.show(False) is scripts.AlwaysVisible and tv.show(True) is scripts.AlwaysVisible


def test_process_argument_order_matches_upstream(plugin):
    # upstream scripts/tilediffusion.py:257-266 and scripts/tilevae.py:704-708 (gradio passes values positionally)
    td = list(inspect.signature(plugin.tilediffusion.Script.process).parameters)
    assert td == ["self", "p", "enabled", "method", "overwrite_size", "keep_input_size", "image_width", "image_height",
                  "tile_width", "tile_height", "overlap", "tile_batch_size", "upscaler_name", "scale_factor",
                  "noise_inverse", "noise_inverse_steps", "noise_inverse_retouch", "noise_inverse_renoise_strength",
                  "noise_inverse_renoise_kernel", "control_tensor_cpu", "enable_bbox_control", "draw_background",
                  "causal_layers", "bbox_control_states"]
    tv = list(inspect.signature(plugin.tilevae.Script.process).parameters)
    assert tv == ["self", "p", "enabled", "encoder_tile_size", "decoder_tile_size", "vae_to_gpu", "fast_decoder",
                  "fast_encoder", "color_fix"]


def test_delegate_method_names(plugin):
    md, mod = plugin.multidiffusion.MultiDiffusion, plugin.mixtureofdiffusers.MixtureOfDiffusers
    for name in ("hook", "unhook", "kdiff_forward", "ddim_forward", "sample_one_step", "repeat_tensor", "repeat_cond_dict",
                 "get_noise", "init_grid_bbox", "init_custom_bbox", "init_done", "reset_buffer", "get_tile_weights"):
        assert callable(getattr(md, name)), name
    for name in ("hook", "unhook", "apply_model_hijack", "custom_apply_model", "get_noise", "init_done", "get_tile_weights"):
        assert callable(getattr(mod, name)), name
    vh = plugin.tilevae.VAEHook
    for name in ("__call__", "split_tiles", "get_best_tile_size", "estimate_group_norm", "vae_tile_forward"):
        assert callable(getattr(vh, name)), name
    assert list(inspect.signature(vh.__init__).parameters) == ["self", "net", "tile_size", "is_decoder", "fast_decoder",
                                                                "fast_encoder", "color_fix", "to_gpu"]


def test_bbox_settings_and_splitable(plugin):
    U = plugin.utils
    assert U.NUM_BBOX_PARAMS == 10 and U.BBoxSettings._fields[0] == "enable" and U.BBoxSettings._fields[-1] == "seed"
    states = [True, 0.123456, 0.2, 0.30004, 0.5, "a", "b", "Background", 0.21234, 7.0,
              False, 0.1, 0.1, 0.1, 0.1, "", "", "Background", 0.2, -1,
              True, 1.5, 0.1, 0.1, 0.1, "", "", "Foreground", 0.2, -1]
    s = U.build_bbox_settings(states)
    assert list(s) == [0] and s[0].x == 0.1235 and s[0].w == 0.3 and s[0].feather_ratio == 0.2123 and s[0].seed == 7
    assert U.splitable(2048, 2048, 96, 96, 48) and not U.splitable(512, 512, 96, 96, 48)
    assert U.Method.MULTI_DIFF == "MultiDiffusion" and U.Method("Mixture of Diffusers") == U.Method.MIX_DIFF
    b = U.BBox(3, 5, 10, 20)
    assert b.box == [3, 5, 13, 25] and b.slicer[2] == slice(5, 25) and b[2] == 13


def test_tilediffusion_process_arms_and_restores_hijack(plugin):
    import modules.sd_samplers as sd_samplers
    orig = sd_samplers.create_sampler
    s = plugin.tilediffusion.Script()
    p = sh.make_processing(2048, 2048)
    p.extra_generation_params = {}
    defaults = list(plugin.utils.DEFAULT_BBOX_SETTINGS) * 8
    s.process(p, True, "MultiDiffusion", False, True, 1024, 1024, 96, 96, 48, 4, "None", 2.0, False, 10, 1, 1, 64, False,
              False, False, False, *defaults)
    assert sd_samplers.create_sampler is not orig and "Tiled Diffusion" in p.extra_generation_params
    s.postprocess(p, None, True)
    assert sd_samplers.create_sampler is orig
    # a canvas that fits one tile arms nothing
    p2 = sh.make_processing(512, 512)
    p2.extra_generation_params = {}
    s.process(p2, True, "MultiDiffusion", False, True, 1024, 1024, 96, 96, 48, 4, "None", 2.0, False, 10, 1, 1, 64, False,
              False, False, False, *defaults)
    assert sd_samplers.create_sampler is orig


def test_tilediffusion_region_control_hijacks_and_restores_random_tensors(plugin):
    """Region control arms the per-region noise hijack (upstream :376-383) and reset() puts the host's function back."""
    import modules.processing as processing
    sentinel = lambda *a, **k: None   # noqa: E731
    old = processing.create_random_tensors
    processing.create_random_tensors = sentinel
    try:
        s = plugin.tilediffusion.Script()
        p = sh.make_processing(2048, 2048)
        p.extra_generation_params = {}
        regions = [True, 0.1, 0.1, 0.4, 0.4, "a cat", "", "Background", 0.2, 42] + list(plugin.utils.DEFAULT_BBOX_SETTINGS) * 7
        s.process(p, True, "MultiDiffusion", False, True, 1024, 1024, 96, 96, 48, 4, "None", 2.0, False, 10, 1, 1, 64, False,
                  True, True, False, *regions)
        assert processing.create_random_tensors is not sentinel
        assert plugin.tilediffusion.Script.create_random_tensors_original_md is sentinel
        assert "Region 1" in p.extra_generation_params["Tiled Diffusion"]["Region control"]
        s.postprocess(p, None, True)
        assert processing.create_random_tensors is sentinel
        assert not hasattr(plugin.tilediffusion.Script, "create_random_tensors_original_md")
    finally:
        processing.create_random_tensors = old


def test_tilevae_process_hooks_both_directions_and_restores(plugin):
    """Script.process replaces encoder.forward AND decoder.forward with VAEHooks (upstream tilevae.py:739-745);
    disabling or postprocess restores the originals."""
    from types import SimpleNamespace
    import torch

    class Net(torch.nn.Module):
        def forward(self, x):
            return x

    enc, dec = Net(), Net()
    enc_fwd, dec_fwd = enc.forward, dec.forward
    p = SimpleNamespace(sd_model=SimpleNamespace(first_stage_model=SimpleNamespace(encoder=enc, decoder=dec, device="cpu")))
    s = plugin.tilevae.Script()
    s.process(p, True, 3072, 256, True, True, True, False)
    VAEHook = plugin.tilevae.VAEHook
    assert isinstance(enc.forward, VAEHook) and isinstance(dec.forward, VAEHook)
    assert (enc.forward.is_decoder, enc.forward.pad, enc.forward.tile_size) == (False, 32, 3072)
    assert (dec.forward.is_decoder, dec.forward.pad, dec.forward.tile_size) == (True, 11, 256)
    assert enc.original_forward == enc_fwd and dec.original_forward == dec_fwd
    s.postprocess(p, None, True)
    assert enc.forward == enc_fwd and dec.forward == dec_fwd
    s.process(p, True, 3072, 256, True, True, True, True)
    assert enc.forward.color_fix and not dec.forward.color_fix          # color_fix is an encoder-only mode (upstream :370)
    s.process(p, False, 3072, 256, True, True, True, False)              # "disabled": undo a hook left over from a crashed job
    assert enc.forward == enc_fwd and dec.forward == dec_fwd


def test_demofusion_script_surface(plugin):
    """scripts/tileglobal.py: title, positional argument order of process (upstream :127-136), delegate method names, hijack restore."""
    import modules.sd_samplers as sd_samplers
    import modules.processing as processing
    s = plugin.tileglobal.Script()
    assert s.title() == "demofusion"
    args = list(inspect.signature(plugin.tileglobal.Script.process).parameters)
    assert args == ["self", "p", "enabled", "method", "keep_input_size", "window_size", "overlap", "tile_batch_size", "scale_factor",
                    "noise_inverse", "noise_inverse_steps", "noise_inverse_retouch", "noise_inverse_renoise_strength",
                    "noise_inverse_renoise_kernel", "control_tensor_cpu", "random_jitter", "c1", "c2", "c3", "gaussian_filter",
                    "strength", "sigma", "batch_size_g", "mixture_mode"]
    df = plugin.demofusion.DemoFusion
    for name in ("hook", "unhook", "forward_one_step", "sample_one_step", "get_views", "split_bboxes_jitter", "global_split_bboxes",
                 "gaussian_kernel", "gaussian_filter", "repeat_tensor", "repeat_cond_dict", "apply_model_hijack", "get_noise"):
        assert callable(getattr(df, name)), name
    orig_cs, orig_ci = sd_samplers.create_sampler, getattr(processing, "create_infotext", None)
    processing.create_infotext = orig_ci or (lambda *a, **k: "")
    keep = processing.create_infotext
    p = sh.make_processing(1024, 1024)
    p.extra_generation_params = {}
    s.process(p, True, "DemoFusion", False, 64, 32, 4, 2.0, False, 10, 1.0, 1.0, 64, False, True, 3, 1, 1, True, 0.85, 0.6, 4, False)
    assert sd_samplers.create_sampler is not orig_cs and processing.create_infotext is not keep and callable(p.sample)
    assert p.extra_generation_params["Tiled Diffusion"]["Method"] == "DemoFusion" and p.scale_factor == 2 and p.mixture is False
    s.postprocess(p, None, True)
    assert sd_samplers.create_sampler is orig_cs and processing.create_infotext is keep
    if orig_ci is None:
        del processing.create_infotext
