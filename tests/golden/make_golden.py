#!/usr/bin/env python3
"""
Regenerates the golden fixtures in this directory by running the UPSTREAM reference code itself
(/root/reference, imported verbatim under hostsim/stub_host.py) on CPU, fp32.

    python tests/golden/make_golden.py

The reference ships no golden vectors of its own; these files are what pins the oracle and
the HIP path on the GPU box, where /root/reference does not exist.  Inputs are never stored -- they are
re-derived from the seeds recorded in cases.json (torch CPU RNG is deterministic for a fixed torch build).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from hostsim import stub_host as sh  # noqa: E402
from oracle import blend_oracle as bo  # noqa: E402  (only for the synthetic denoisers + region_rect helper)
from hostsim import ldm_decoder as ld  # noqa: E402

REGIONS_CFG5 = [  # fractions of the canvas, as in BBoxSettings
    (0.0, 0.0, 0.4, 1.0, "Background", 0.2),
    (0.3, 0.0, 0.4, 1.0, "Background", 0.2),
    (0.6, 0.1, 0.4, 0.8, "Foreground", 0.2),
]
REGIONS_2FG = REGIONS_CFG5 + [(0.5, 0.2, 0.3, 0.6, "Foreground", 0.5)]

BLEND_CASES = [
    # name, method, W, H, tw, th, overlap, tile_bs, regions, draw_background, N, seed
    dict(name="md_cfg2_small", method="md", W=96, H=96, tw=36, th=36, ov=18, bs=4, regions=None, bg=True, N=2, seed=0),
    dict(name="md_ragged", method="md", W=100, H=70, tw=32, th=24, ov=8, bs=3, regions=None, bg=True, N=2, seed=1),
    dict(name="md_cfg5_regions", method="md", W=128, H=32, tw=24, th=24, ov=12, bs=4, regions=REGIONS_CFG5, bg=True, N=2, seed=2),
    dict(name="md_nobg_2fg", method="md", W=64, H=48, tw=16, th=16, ov=4, bs=4, regions=REGIONS_2FG, bg=False, N=2, seed=3),
    dict(name="md_single_tile", method="md", W=40, H=40, tw=96, th=96, ov=48, bs=4, regions=REGIONS_CFG5, bg=True, N=1, seed=4),
    dict(name="mod_cfg3_small", method="mod", W=96, H=96, tw=36, th=36, ov=18, bs=4, regions=None, bg=True, N=2, seed=5),
    dict(name="mod_ragged", method="mod", W=100, H=70, tw=32, th=24, ov=8, bs=3, regions=None, bg=True, N=3, seed=6),
    dict(name="mod_cfg5_regions", method="mod", W=128, H=32, tw=24, th=24, ov=12, bs=4, regions=REGIONS_CFG5, bg=True, N=2, seed=7),
    dict(name="mod_nobg_2fg", method="mod", W=64, H=48, tw=16, th=16, ov=4, bs=4, regions=REGIONS_2FG, bg=False, N=2, seed=8),
]

GRID_CASES = [  # (w, h, tile_w, tile_h, overlap, tile_bs) -- BASELINE configs + ragged ones
    (256, 256, 96, 96, 48, 4), (512, 512, 96, 96, 48, 4), (512, 512, 96, 96, 8, 4), (1024, 1024, 128, 128, 8, 4),
    (1024, 1024, 128, 128, 16, 4), (1024, 1024, 128, 128, 64, 4), (512, 128, 96, 96, 48, 4), (64, 64, 96, 96, 48, 4),
    (100, 70, 32, 24, 8, 3), (97, 131, 16, 16, 15, 8), (97, 131, 16, 16, 200, 8), (33, 200, 40, 40, 0, 1),
]

TILE_CASES = [  # (h, w, tile_size, is_decoder)
    (64, 64, 64, True), (256, 256, 64, True), (512, 512, 64, True), (512, 512, 256, True), (1024, 1024, 64, True),
    (1024, 1024, 256, True), (128, 512, 64, True), (128, 512, 256, True), (40, 56, 16, True), (30, 70, 24, True),
    (300, 23, 48, True), (2048, 2048, 512, False), (4096, 1024, 960, False), (1000, 777, 512, False),
]

VAE_CASES = [  # small-decoder (same topology, ch=32) tiled decodes
    dict(name="vae_fast", H=36, W=44, ts=16, fast=True, seed=2, dec_seed=0),
    dict(name="vae_slow", H=36, W=44, ts=16, fast=False, seed=2, dec_seed=0),
    dict(name="vae_fast_ragged", H=30, W=70, ts=24, fast=True, seed=3, dec_seed=1),
]
VAE_STRIDE = 3


def run_ref_blend(ref, c):
    dev, shared = sh.host()
    U = ref.utils
    p = sh.make_processing(c["W"] * 8, c["H"] * 8)
    cls = ref.multidiffusion.MultiDiffusion if c["method"] == "md" else ref.mixtureofdiffusers.MixtureOfDiffusers
    d = cls(p, sh.kdiff_sampler())
    d.init_grid_bbox(c["tw"], c["th"], c["ov"], c["bs"])
    if c["regions"]:
        # init_custom_bbox (abstractdiffusion.py:194-229) minus the prompt/cond machinery, which needs a real host
        d.enable_custom_bbox = True
        d.draw_background = c["bg"]
        if not c["bg"]:
            d.enable_grid_bbox = False
            d.weights.zero_()
        d.custom_bboxes = []
        for (fx, fy, fw, fh, mode, fr) in c["regions"]:
            x, y, w, h = bo.region_rect(c["W"], c["H"], fx, fy, fw, fh)
            d.custom_bboxes.append(U.CustomBBox(x, y, w, h, "", "", mode, fr, -1))
        for b in d.custom_bboxes:  # the subclass halves of init_custom_bbox (multidiffusion.py:44-46, mixtureofdiffusers.py:49-55)
            if c["method"] == "md":
                if b.blend_mode == U.BlendMode.BACKGROUND:
                    d.weights[b.slicer] += 1.0
            else:
                if b.blend_mode == U.BlendMode.BACKGROUND:
                    cw = d.get_weight(b.w, b.h)
                    d.weights[b.slicer] += cw
                    d.custom_weights.append(cw.unsqueeze(0).unsqueeze(0))
                else:
                    d.custom_weights.append(None)
    d.init_done()
    d.update_pbar = lambda: None
    d.pbar.close()
    torch.manual_seed(c["seed"])
    x = torch.randn(c["N"], 4, c["H"], c["W"])
    if c["method"] == "md":
        out = d.sample_one_step(x, None, lambda xt, b: bo.synthetic_denoiser(xt),
                                lambda xr, i, b: bo.synthetic_region_denoiser(xr, i))
    else:
        shared.sd_model.apply_model_original_md = lambda x_, t_, c_: bo.synthetic_denoiser(x_)
        d.custom_apply_model = lambda x_in, t_in, c_in, bbox_id, bbox: bo.synthetic_region_denoiser(x_in, bbox_id)
        cond = {"c_crossattn": [torch.zeros(c["N"], 77, 768)], "c_concat": [torch.zeros(c["N"], 5, 1, 1)]}
        out = d.apply_model_hijack(x, torch.zeros(c["N"]), cond)
    return out, d.weights


def main():
    ref = sh.load_reference()
    cases = {"blend": BLEND_CASES, "grid": [], "tiles": [], "vae": VAE_CASES, "vae_stride": VAE_STRIDE}

    blend = {}
    for c in BLEND_CASES:
        out, weights = run_ref_blend(ref, c)
        blend[c["name"] + "/out"] = out.numpy()
        blend[c["name"] + "/weights"] = weights.numpy()
    np.savez_compressed(os.path.join(HERE, "blend.npz"), **blend)

    maps = {}
    for tw, th in [(96, 96), (128, 128), (36, 36), (32, 24), (24, 24), (16, 16), (33, 57), (52, 32)]:
        maps[f"gauss_{tw}x{th}"] = ref.utils.gaussian_weights(tw, th).numpy()
    for w, h, r in [(40, 30, 0.2), (41, 33, 0.5), (10, 10, 0.0), (64, 64, 1.0), (52, 26, 0.2), (7, 9, 0.9)]:
        maps[f"feather_{w}x{h}_{r}"] = ref.utils.feather_mask(w, h, r).numpy()
    np.savez_compressed(os.path.join(HERE, "maps.npz"), **maps)

    for (w, h, tw, th, ov, bs) in GRID_CASES:
        p = sh.make_processing(w * 8, h * 8)
        d = ref.multidiffusion.MultiDiffusion(p, sh.kdiff_sampler())
        d.init_grid_bbox(tw, th, ov, bs)
        boxes = [[b.x, b.y, b.w, b.h] for batch in d.batched_bboxes for b in batch]
        cases["grid"].append(dict(args=[w, h, tw, th, ov, bs], boxes=boxes, num_batches=d.num_batches, tile_bs=d.tile_bs,
                                  wmin=float(d.weights.min()), wmax=float(d.weights.max()), wsum=float(d.weights.double().sum())))

    for (h, w, ts, is_dec) in TILE_CASES:
        hook = ref.tilevae.VAEHook(None, ts, is_decoder=is_dec, fast_decoder=True, fast_encoder=True, color_fix=False)
        ins, outs = hook.split_tiles(h, w)
        cases["tiles"].append(dict(args=[h, w, ts, is_dec], ins=ins, outs=outs))

    vae = {}
    for c in VAE_CASES:
        dec = ld.make_decoder(c["dec_seed"], small=True)
        dec.original_forward = dec.forward
        torch.manual_seed(c["seed"])
        z = torch.randn(1, 4, c["H"], c["W"])
        hook = ref.tilevae.VAEHook(dec, c["ts"], is_decoder=True, fast_decoder=c["fast"], fast_encoder=False, color_fix=False)
        out = hook(z)
        vae[c["name"] + "/sub"] = out[:, :, ::VAE_STRIDE, ::VAE_STRIDE].contiguous().numpy()
        vae[c["name"] + "/moments"] = np.array([out.double().sum().item(), (out.double() ** 2).sum().item(),
                                                out.abs().max().item()], dtype=np.float64)
    # GroupNorm primitives on a fixed tensor (tilevae.py:207-245)
    torch.manual_seed(11)
    t = torch.randn(2, 64, 9, 13) * 3 + 0.5
    var, mean = ref.tilevae.get_var_mean(t, 32)
    g = torch.randn(64)
    b = torch.randn(64)
    vae["gn/var"], vae["gn/mean"] = var.numpy(), mean.numpy()
    vae["gn/out"] = ref.tilevae.custom_group_norm(t, 32, mean, var, g, b).numpy()
    # attention body (attn.py:49-72) on a small AttnBlock
    torch.manual_seed(12)
    ab = ld.AttnBlock(64).eval()
    hx = torch.randn(1, 64, 7, 9)
    with torch.no_grad():
        vae["attn/out"] = ref.attn.attn_forward(ab, hx).numpy()
    np.savez_compressed(os.path.join(HERE, "vae.npz"), **vae)

    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    for fn in ("blend.npz", "maps.npz", "vae.npz", "cases.json"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
