"""Shared driver of the MultiDiffusion ENTRY-path tests (test infrastructure): builds a delegate (upstream's or this repo's), calls
`hook()`, then drives `sampler.model_wrap_cfg.inner_model.forward(x, sigma_or_ts, cond=...)` for a few sampler steps -- the call
A1111's CFG denoiser makes (SURVEY.md section 8b "Hijack points") -- with a stand-in UNet whose output depends on the tile content,
sigma / timestep, the text conditioning rows, the image conditioning (latent-sized => sliced per bbox) and SDXL's vector
conditioning.  Every operation of the stand-in is an elementwise IEEE op, so CPU and GPU results are bit-identical and any routing
error (tile order, cond row order, icond slicing, sigma repeat) changes the output."""
from __future__ import annotations

import sys
from types import SimpleNamespace

import torch

from oracle import blend_oracle as bo
from oracle import entry_oracle as eo
from hostsim import stub_host as sh

REGIONS = [  # fractions of the canvas, BBoxSettings style
    (0.0, 0.0, 0.45, 1.0, "Background", 0.2),
    (0.3, 0.1, 0.5, 0.8, "Foreground", 0.3),
    (0.55, 0.0, 0.45, 0.9, "Background", 0.2),
]

ENTRY_CASES = [
    # name, sampler, cond style, img2img icond?, regions?, N, W, H, tw, th, ov, bs
    dict(name="kdiff_sd1_t2i", sampler="kdiff", style="sd1", i2i=False, regions=False, N=2, W=56, H=40, tw=24, th=16, ov=8, bs=3),
    dict(name="kdiff_sd1_i2i", sampler="kdiff", style="sd1", i2i=True, regions=False, N=2, W=56, H=40, tw=24, th=16, ov=8, bs=3),
    dict(name="kdiff_sdxl_i2i", sampler="kdiff", style="sdxl", i2i=True, regions=False, N=2, W=50, H=36, tw=16, th=16, ov=4, bs=4),
    dict(name="kdiff_sd1_i2i_regions", sampler="kdiff", style="sd1", i2i=True, regions=True, N=2, W=56, H=40, tw=24, th=16, ov=8, bs=3),
    dict(name="kdiff_batch1", sampler="kdiff", style="sd1", i2i=True, regions=False, N=1, W=40, H=40, tw=16, th=24, ov=8, bs=2),
    dict(name="kdiff_batch2_sdxl", sampler="kdiff", style="sdxl", i2i=True, regions=False, N=4, W=48, H=48, tw=24, th=24, ov=12, bs=4),   # batch 2 x (cond, uncond)
    dict(name="kdiff_single_tile_batches", sampler="kdiff", style="sd1", i2i=True, regions=False, N=2, W=56, H=40, tw=24, th=16, ov=8, bs=1),   # tile batch 1: repeat_tensor's n == 1 path
    dict(name="ddim_dict_i2i", sampler="ddim", style="sd1", i2i=True, regions=False, N=2, W=56, H=40, tw=24, th=16, ov=8, bs=3),
    dict(name="ddim_tensor_cond", sampler="ddim", style="tensor", i2i=False, regions=False, N=2, W=56, H=40, tw=24, th=16, ov=8, bs=3),
    dict(name="ddim_dict_i2i_regions", sampler="ddim", style="sd1", i2i=True, regions=True, N=1, W=56, H=40, tw=24, th=16, ov=8, bs=3),
]
STEPS = 3
SIGMAS = [7.5, 3.25, 0.75]


def _t(cond):
    if not isinstance(cond, dict):
        return cond
    t = cond["crossattn" if "crossattn" in cond else "c_crossattn"]
    return t[0] if isinstance(t, list) else t


def _core(x, sig, cond):
    t = _t(cond)
    assert t.shape[0] == x.shape[0] == sig.shape[0], f"batch mismatch: x {x.shape[0]} cond {t.shape[0]} sigma {sig.shape[0]}"
    out = x * 0.5 + x.flip(-1) * 0.25 + sig.view(-1, 1, 1, 1) * 0.125 + t[:, 0, 0].view(-1, 1, 1, 1)
    if isinstance(cond, dict):
        i = cond["c_concat"]
        i = i[0] if isinstance(i, list) else i
        assert i.shape[0] == x.shape[0], f"icond rows {i.shape[0]} vs x rows {x.shape[0]}"
        out = out + i[:, :4] * 0.0625          # [B,4,th,tw] (img2img, sliced per tile) or [B,4,1,1] (txt2img dummy, broadcast)
        v = cond.get("vector")
        if v is not None:
            assert v.shape[0] == x.shape[0]
            out = out + v[:, 1].view(-1, 1, 1, 1)
    return out


def make_model(calls):
    """forward(x, sigma_or_ts, cond=...) -- plus the legacy CompVis form forward(x, c, ts, unconditional_conditioning=uc) that
    ddim_custom_forward uses for a region (abstractdiffusion.py:451)."""
    def forward(x, a, b=None, cond=None, unconditional_conditioning=None):
        if cond is None:
            c, ts, uc = a, b, unconditional_conditioning
            calls.append(("region", x.shape[0], _t(c).shape[1]))
            return _core(x, ts, c) + _t(uc)[:, 0, 1].view(-1, 1, 1, 1) * 0.5
        calls.append(("model", x.shape[0], _t(cond).shape[1]))
        return _core(x, a, cond)
    return forward


def inputs(case, device="cpu"):
    g = torch.Generator().manual_seed(17)
    N, W, H = case["N"], case["W"], case["H"]
    x = torch.randn(N, 4, H, W, generator=g)
    tc = torch.randn(N, 77, 8, generator=g)
    ic = torch.randn(N, 5, H, W, generator=g) if case["i2i"] else torch.randn(N, 5, 1, 1, generator=g)
    vc = torch.randn(N, 16, generator=g)
    x, tc, ic, vc = (t.to(device) for t in (x, tc, ic, vc))
    if case["style"] == "sd1":
        cond = {"c_crossattn": [tc], "c_concat": [ic]}
    elif case["style"] == "sdxl":
        cond = {"crossattn": tc, "vector": vc, "c_concat": [ic]}
    else:
        cond = tc
    tens = {}
    for i in range(len(REGIONS)):
        tens[f"rc{i}"] = torch.randn(1, 77, 8, generator=g).to(device)
        tens[f"ru{i}"] = torch.randn(1, 77 if i != 1 else 154, 8, generator=g).to(device)     # region 1: negative prompt > 75 tokens
    tens["gc"] = torch.randn(1, 77, 8, generator=g).to(device)
    tens["gu"] = torch.randn(1, 77, 8, generator=g).to(device)
    return x, cond, tens


def region_tensor(tens, name, step):
    return tens[name] + float(step)            # step-dependent, like a scheduled prompt


def make_sampler(kind):
    if kind == "kdiff":
        smp = sh.kdiff_sampler()
    else:
        smp = sys.modules["modules.sd_samplers_timesteps"].CompVisSampler()
    smp.model_wrap_cfg = SimpleNamespace(step=0, inner_model=SimpleNamespace(forward=None), image_cfg_scale=None)
    return smp


def drive(mods, case, device="cpu", build_regions=None):
    """Run STEPS sampler steps through the hooked inner model of `mods` (reference or plugin namespace).  Returns (outputs, calls)."""
    x, cond, tens = inputs(case, device)
    if case["sampler"] == "ddim" and case["regions"]:
        # a CompVis sampler hands cond and uncond separately: a region's uncond is padded / truncated to the prompt's length
        # (abstractdiffusion.py:436-441), which the kdiff whole-batch branch cannot take -- keep region 1's long negative prompt there
        pass
    else:
        tens["ru1"] = tens["ru1"][:, :77].contiguous()
    C = mods.utils.Condition
    old = (C.reconstruct_cond, C.reconstruct_uncond)
    C.reconstruct_cond = staticmethod(lambda c, step: region_tensor(tens, c, step))
    C.reconstruct_uncond = staticmethod(lambda c, step: region_tensor(tens, c, step))
    calls = []
    try:
        smp = make_sampler(case["sampler"])
        smp.model_wrap_cfg.inner_model.forward = make_model(calls)
        p = sh.make_processing(case["W"] * 8, case["H"] * 8, sampler_name="Euler" if case["sampler"] == "kdiff" else "DDIM")
        d = mods.multidiffusion.MultiDiffusion(p, smp)
        d.init_grid_bbox(case["tw"], case["th"], case["ov"], case["bs"])
        if case["regions"]:
            build_regions(d, case)
            for i, b in enumerate(d.custom_bboxes):
                b.cond, b.uncond = f"rc{i}", f"ru{i}"
            d.cond_basis, d.uncond_basis = "gc", "gu"
        d.init_done()
        if d.pbar is not None:
            d.pbar.close()
        d.update_pbar = lambda: None
        d.hook()
        assert smp.model_wrap_cfg.inner_model.forward != d.sampler_forward, "hook() did not replace inner_model.forward"
        outs = []
        for k in range(STEPS):
            smp.model_wrap_cfg.step = k
            sig = torch.full((case["N"],), SIGMAS[k], device=device)
            out = smp.model_wrap_cfg.inner_model.forward(x, sig, cond=cond)
            outs.append(out)
            x = x - out * 0.25
        # hires second pass: a latent of another spatial size bypasses the tiling entirely (multidiffusion.py:140-144) -- one model call
        # with the sampler's own arguments
        if not case["i2i"] and not case["regions"]:
            n0 = len(calls)
            xs = x[:, :, :case["H"] // 2, :case["W"] // 2].contiguous()
            sig = torch.full((case["N"],), 0.5, device=device)
            direct = make_model([])(xs, sig, cond=cond)
            byp = smp.model_wrap_cfg.inner_model.forward(xs, sig, cond=cond)
            assert len(calls) == n0 + 1 and torch.equal(byp, direct), "size mismatch must fall through to the original forward"
            del calls[n0:]
        return torch.stack(outs), calls
    finally:
        C.reconstruct_cond, C.reconstruct_uncond = old


def ref_regions(ref):
    """Upstream delegate: init_custom_bbox minus the prompt parser (as tests/golden/make_golden.py does)."""
    def build(d, case):
        U = ref.utils
        d.enable_custom_bbox = True
        d.draw_background = True
        d.custom_bboxes = []
        for (fx, fy, fw, fh, mode, fr) in REGIONS:
            x, y, w, h = bo.region_rect(case["W"], case["H"], fx, fy, fw, fh)
            d.custom_bboxes.append(U.CustomBBox(x, y, w, h, "", "", mode, fr, -1))
        for b in d.custom_bboxes:
            if b.blend_mode == U.BlendMode.BACKGROUND:
                d.weights[b.slicer] += 1.0
    return build


def plugin_regions(plugin):
    def build(d, case):
        U = plugin.utils
        settings = {i: U.BBoxSettings(True, fx, fy, fw, fh, "", "", mode, fr, -1) for i, (fx, fy, fw, fh, mode, fr) in enumerate(REGIONS)}
        d.init_custom_bbox(settings, True, False)
    return build


def oracle_run(case):
    """The same three steps on the restatement (oracle/entry_oracle.py), CPU."""
    x, cond, tens = inputs(case, "cpu")
    if not (case["sampler"] == "ddim" and case["regions"]):
        tens["ru1"] = tens["ru1"][:, :77].contiguous()
    regs = [bo.Region(*bo.region_rect(case["W"], case["H"], fx, fy, fw, fh), mode, fr) for (fx, fy, fw, fh, mode, fr) in REGIONS] if case["regions"] else []
    o = bo.BlendOracle("md", case["W"], case["H"], case["tw"], case["th"], case["ov"], case["bs"], regs, True)
    rconds = [(lambda s, i=i: region_tensor(tens, f"rc{i}", s), lambda s, i=i: region_tensor(tens, f"ru{i}", s)) for i in range(len(regs))]
    calls = []
    forward = make_model(calls)
    fn = eo.kdiff_forward if case["sampler"] == "kdiff" else eo.ddim_forward
    outs = []
    for k in range(STEPS):
        sig = torch.full((case["N"],), SIGMAS[k])
        out = fn(o, x, sig, cond, forward, k, rconds)
        outs.append(out)
        x = x - out * 0.25
    return torch.stack(outs), calls
