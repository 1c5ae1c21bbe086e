"""GPU parity of the SURVEY section 8f rows built on the engine: Noise Inversion's renoise composite (mdtile_noise_inverse_blend,
upstream abstractdiffusion.py:651-676) and ControlNet / StableSR tile slicing (mdtile_gather_rects, :475-544, :548-588), both at
the kernel level (bit-exact against the oracle / against plain slicing) and through the delegates of the plugin surface."""
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F

from oracle import blend_oracle as bo
from hostsim import stub_host as sh

pytestmark = pytest.mark.gpu

NI_REGIONS = [(3, 2, 20, 12, "Background", 0.2), (10, 6, 18, 14, "Foreground", 0.3), (16, 10, 20, 12, "Foreground", 0.6),
              (0, 14, 9, 10, "Background", 0.2)]


def _ni_case(W=40, H=28, seed=7):
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(2, 4, H, W, generator=g)
    inverse = torch.randn(2, 4, H, W, generator=g) * 3.0
    mask = torch.clamp(torch.rand(H, W, generator=g) * 1.4 - 0.2, 0, 1)
    return noise, inverse, mask


@pytest.mark.parametrize("grid", [True, False])
def test_noise_inverse_blend_bit_exact(plugin, cuda, grid):
    E = plugin.engine
    noise, inverse, mask = _ni_case()
    regs = [bo.Region(*r) for r in NI_REGIONS]
    ref = bo.noise_inverse_blend(noise, inverse, mask, regs, enable_grid_bbox=grid)
    eng_regs = []
    if not grid:
        for (x, y, w, h, mode, fr) in NI_REGIONS:
            fg = mode == "Foreground"
            eng_regs.append((x, y, w, h, E.REGION_FG if fg else E.REGION_BG, E.feather_mask(w, h, fr, cuda) if fg else None))
    out = E.noise_inverse_blend(noise.to(cuda), inverse.to(cuda), mask.to(cuda), eng_regs).cpu()
    # Every op of the composite is a correctly rounded IEEE fp32 op in the kernel.  torch's CPU sqrt is NOT (its vectorised sqrt is
    # off by one ulp on ~1 % of inputs), so the oracle -- pinned bit-exact to upstream ON THE CPU -- can differ in the last bit of the
    # denominator: compare exactly against the same op sequence in numpy (IEEE sqrt), and within 2 ulp against the oracle.
    import numpy as np
    n_ = (ref_noise_layer(noise, regs) if not grid else noise).numpy()
    m_ = mask.numpy()
    om = np.float32(1) - m_
    exact = (om * inverse.numpy() + m_ * n_) / np.sqrt(m_ * m_ + om * om)
    assert np.array_equal(out.numpy(), exact)
    assert torch.allclose(out, ref, rtol=3e-7, atol=0)


def ref_noise_layer(noise, regs):
    """The region re-weighting of the job's noise (upstream :658-672) through the oracle: with an all-zero mask the composite
    degenerates to ... the inverse noise, so recover the layer by calling the oracle with mask 1 (combined = noise' / 1)."""
    H, W = noise.shape[2], noise.shape[3]
    return bo.noise_inverse_blend(noise, torch.zeros_like(noise), torch.ones(H, W), regs, enable_grid_bbox=False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_gather_rects_equals_slicing(plugin, cuda, dtype):
    E = plugin.engine
    torch.manual_seed(3)
    x = torch.randn(1, 3, 40 * 8, 56 * 8).to(dtype)
    rects = [(0, 0), (16, 8), (40, 24), (8, 16)]          # latent-grid origins of 16 x 16 tiles
    w = h = 16 * 8
    rows = torch.cat([x[:, :, y * 8:y * 8 + h, xx * 8:xx * 8 + w] for (xx, y) in rects], dim=0)
    # k-diffusion: every tile's copies consecutive (upstream :528-533)
    want = torch.cat([rows[i:i + 1].repeat(3, 1, 1, 1) for i in range(len(rects))], dim=0)
    got = E.gather_rects(x.to(cuda), [(xx * 8, y * 8) for (xx, y) in rects], w, h, repeat=3, tile_major=True)
    assert torch.equal(got.cpu(), want)
    # DDIM: the whole stack repeated (upstream :535)
    got = E.gather_rects(x.to(cuda), [(xx * 8, y * 8) for (xx, y) in rects], w, h, repeat=2, tile_major=False)
    assert torch.equal(got.cpu(), rows.repeat(2, 1, 1, 1))
    # several samples (StableSR latent image, scale 1): torch.cat order = tile-major, then samples
    z = torch.randn(2, 4, 40, 56).to(dtype)
    got = E.gather_rects(z.to(cuda), rects, 16, 16)
    assert torch.equal(got.cpu(), torch.cat([z[:, :, y:y + 16, xx:xx + 16] for (xx, y) in rects], dim=0))


def _delegate(plugin, cuda, W, H, kdiff=True):
    p = sh.make_processing(W * 8, H * 8)
    smp = sh.kdiff_sampler() if kdiff else __import__("sys").modules["modules.sd_samplers_timesteps"].CompVisSampler()
    smp.model_wrap_cfg = SimpleNamespace(step=0, inner_model=SimpleNamespace(forward=None), image_cfg_scale=None)
    cls = plugin.multidiffusion.MultiDiffusion
    cls.is_edit_model = False
    d = cls(p, smp)
    return d, p


@pytest.mark.parametrize("kdiff", [True, False])
def test_controlnet_and_stablesr_tensors_follow_the_tiles(plugin, cuda, kdiff):
    W, H = 56, 40
    d, p = _delegate(plugin, cuda, W, H, kdiff)
    d.init_grid_bbox(16, 16, 4, 3)
    d.custom_bboxes = [plugin.utils.CustomBBox(5, 7, 20, 12, "", "", "Background", 0.2, 1)]
    torch.manual_seed(1)
    hints = [torch.randn(1, 3, H * 8, W * 8, device=cuda), torch.randn(3, H * 8, W * 8, device=cuda)]   # one of them 3-D, as ControlNet stores it
    params = [SimpleNamespace(hint_cond=h) for h in hints]
    script = SimpleNamespace(latest_network=SimpleNamespace(control_params=params))
    d.init_controlnet(script, control_tensor_cpu=False)
    full = [h if h.dim() == 4 else h.unsqueeze(0) for h in hints]
    for batch_id, bboxes in enumerate(d.batched_bboxes):
        d.switch_controlnet_tensors(batch_id, 2, len(bboxes), is_denoise=False)
        for prm, hint in zip(params, full):
            tiles = torch.cat([hint[:, :, b.y * 8:(b.y + b.h) * 8, b.x * 8:(b.x + b.w) * 8] for b in bboxes], dim=0)
            if kdiff:
                want = torch.cat([tiles[i:i + 1].repeat(2, 1, 1, 1) for i in range(len(bboxes))], dim=0)
            else:
                want = tiles.repeat(4, 1, 1, 1)                          # x_batch_size * 2 when not denoising
            assert torch.equal(prm.hint_cond, want)
    d.set_custom_controlnet_tensors(0, 3)
    b = d.custom_bboxes[0]
    for prm, hint in zip(params, full):
        assert torch.equal(prm.hint_cond, hint[:, :, b.y * 8:(b.y + b.h) * 8, b.x * 8:(b.x + b.w) * 8].repeat(3, 1, 1, 1))
    d.reset_controlnet_tensors()
    assert all(torch.equal(prm.hint_cond, h) for prm, h in zip(params, full))
    # StableSR: the latent image is tiled like the latent itself
    model = SimpleNamespace(set_image_hooks={}, latent_image=None)
    d.init_stablesr(SimpleNamespace(stablesr_model=model))
    lat = torch.randn(2, 4, H, W, device=cuda)
    model.set_image_hooks["TiledDiffusion"](lat)
    d.switch_stablesr_tensors(1)
    bb = d.batched_bboxes[1]
    assert torch.equal(model.latent_image, torch.cat([lat[:, :, q.y:q.y + q.h, q.x:q.x + q.w] for q in bb], dim=0))
    d.set_custom_stablesr_tensors(0)
    assert torch.equal(model.latent_image, lat[:, :, b.y:b.y + b.h, b.x:b.x + b.w])
    d.reset_stablesr_tensors()
    assert model.latent_image is lat


@pytest.mark.parametrize("grid", [True, False])
def test_noise_inversion_sample_img2img_through_the_delegate(plugin, cuda, grid, monkeypatch):
    """The product's sample_img2img replacement (cached inversion latent, stubbed retouch mask) hands the ORIGINAL sample_img2img
    the composite upstream computes (:606-681)."""
    from PIL import Image
    import sys
    W, H = 40, 28
    d, p = _delegate(plugin, cuda, W, H)
    if grid:
        d.init_grid_bbox(16, 16, 4, 2)
    d.enable_grid_bbox = grid
    d.custom_bboxes = [plugin.utils.CustomBBox(x, y, w, h, "", "", mode, fr, 1) for (x, y, w, h, mode, fr) in NI_REGIONS]
    g = torch.Generator().manual_seed(11)
    noise = torch.randn(2, 4, H, W, generator=g)
    init_latent = torch.randn(2, 4, H, W, generator=g)
    xt = torch.randn(2, 4, H, W, generator=g) * 3.0
    np_mask = torch.rand(H * 8, W * 8, generator=g).numpy()
    sigmas = torch.linspace(7.5, 0.03, 9)
    p.init_images = [Image.new("RGB", (W * 8, H * 8))]
    p.sd_model = SimpleNamespace(sd_model_hash="hash")
    p.init_latent = init_latent.to(cuda)
    cache = plugin.utils.NoiseInverseCache("hash", init_latent.clone(), xt, 5, 1.0, [""])
    captured = {}
    smp = d.sampler_raw
    smp.sample_img2img = lambda p_, x_, n_, c_, uc_, steps_=None, ic_=None: captured.setdefault("noise", n_)
    d.init_noise_inverse(5, 1.0, lambda: cache, lambda *a: None, 0.7, 3)
    monkeypatch.setattr(plugin.abstractdiffusion, "get_retouch_mask", lambda img, k: np_mask)
    monkeypatch.setattr(sys.modules["modules.sd_samplers_common"], "setup_img2img_steps", lambda p_, steps: (steps or 8, 6), raising=False)
    smp.get_sigmas = lambda p_, steps: sigmas.to(cuda)
    smp.sample_img2img(p, torch.zeros_like(noise).to(cuda), noise.to(cuda), None, None, 8, None)
    m = 1 - F.interpolate(torch.from_numpy(np_mask).unsqueeze(0).unsqueeze(0), size=(H, W), mode="bilinear").squeeze(0).squeeze(0)
    m = torch.clamp(m * 0.7, 0, 1)
    ref = bo.noise_inverse_blend(noise, xt - init_latent / sigmas[0], m, [bo.Region(*r) for r in NI_REGIONS], grid)
    # the bilinear resize of the mask runs on the GPU in the product (host torch op): allow its last-ulp differences
    assert torch.allclose(captured["noise"].cpu(), ref, rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------------
# DemoFusion: the delegate's model evaluation on the engine against the oracle (pinned to upstream)
# ---------------------------------------------------------------------------------------------------------------------
def _demo_tile_fn(x):
    return 0.9 * x + 0.1 * x.flip(-1) + 0.05 * x.flip(-2)


DEMO_CASES = [  # W0, H0, S, window, overlap, jitter, mixture
    (24, 24, 2, 16, 8, True, False),
    (24, 24, 3, 16, 8, True, True),
    (24, 24, 2, 16, 8, False, False),
    (20, 20, 2, 16, 4, True, True),
    (32, 32, 4, 32, 16, True, False),
]


@pytest.mark.parametrize("W0,H0,S,window,overlap,jitter,mixture", DEMO_CASES)
def test_demofusion_sample_one_step_vs_oracle(plugin, cuda, W0, H0, S, window, overlap, jitter, mixture):
    import random
    from oracle import demofusion_oracle as do
    W, H = W0 * S, H0 * S
    p = sh.make_processing(W * 8, H * 8)
    p.random_jitter, p.mixture, p.current_scale_num, p.gaussian_filter = jitter, mixture, S, True
    p.cosine_scale_2, p.cosine_scale_3 = 1.0, 1.0
    p.sd_model = SimpleNamespace(apply_model=lambda x, t, cond: _demo_tile_fn(x))
    smp = sh.kdiff_sampler()
    smp.model_wrap_cfg = SimpleNamespace(step=0, inner_model=SimpleNamespace(forward=None), image_cfg_scale=None, forward=None)
    cls = plugin.demofusion.DemoFusion
    cls.is_edit_model = False
    d = cls(p, smp)
    d.window_size, d.sig = window, 0.3
    d.w, d.h = W, H
    random.seed(1234)
    d.get_views(overlap, 3, 2)
    random.seed(1234)
    origins, J, _, _ = do.views(W, H, window, overlap, jitter)
    assert d.jitter_range == J and [(b.x, b.y) for bb in d.batched_bboxes for b in bb] == origins
    d.sampler_forward = lambda x, sigma, cond: _demo_tile_fn(x)
    d.cosine_factor = 0.5 * (1 + torch.cos(torch.pi * torch.tensor((3 + 1) / (10 + 1))))
    torch.manual_seed(3)
    x = torch.randn(2, 4, H + 2 * J, W + 2 * J)
    cond = {"c_crossattn": [torch.zeros(2, 77, 8, device=cuda)], "c_concat": [torch.zeros(2, 5, 1, 1, device=cuda)]}
    got = d.sample_one_step(x.to(cuda), torch.ones(2, device=cuda), cond).cpu()
    want = do.sample_one_step(x, origins, window, J, 3, 2, S, mixture, True, 0.3, d.cosine_factor, 1.0, 1.0, _demo_tile_fn)
    # the local path and the scatter / mix are the same fp32 operations in the same order; the Gaussian filter's tap order and the
    # std reduction differ from torch's (conv2d / std are not order-specified): fp32 round-off only
    assert torch.allclose(got, want, rtol=2e-5, atol=2e-5), f"max diff {(got - want).abs().max().item()}"


def test_demofusion_local_and_scatter_paths_bit_exact(plugin, cuda):
    """window blend and lattice scatter / mix in isolation: identical to the eager op sequence (fp32, list order)."""
    import random
    from oracle import demofusion_oracle as do
    E = plugin.engine
    W = H = 48
    random.seed(7)
    origins, J, ov, stride = do.views(W, H, 16, 8, True)
    import math
    cols = math.ceil((W - ov) / (16 - ov))
    nom = [min(int(c * ((W - 16) / (cols - 1))), W - 16) for c in range(cols)]
    N, C, Hp, Wp = 2, 4, H + 2 * J, W + 2 * J
    torch.manual_seed(2)
    tiles = torch.randn(len(origins) * N, C, 16, 16)
    buf, cnt = torch.zeros(N, C, Hp, Wp), torch.zeros(N, C, Hp, Wp)
    for i, (x, y) in enumerate(origins):
        buf[:, :, y:y + 16, x:x + 16] += tiles[i * N:(i + 1) * N]
        cnt[:, :, y:y + 16, x:x + 16] += 1
    want = buf / torch.where(cnt == 0, torch.tensor(1), cnt)
    ws = E.WindowSet(origins, nom, nom, J, 16, cuda)
    got = E.window_blend(tiles.to(cuda), ws, N, C, Hp, Wp).cpu()
    assert torch.equal(got, want)
    # lattice gather / scatter + mix
    S = 3
    x = torch.randn(N, C, Hp, Wp)
    xg = torch.randn(N, C, Hp, Wp)
    end = Wp - J
    cells = [(bx, by) for by in range(S) for bx in range(S)]
    h0 = len(range(J, end, S))
    g = E.dilated_gather(x.to(cuda), xg.to(cuda), 4, cells, S, J, h0, h0).cpu()
    ref = torch.cat([(x if i < 4 else xg)[:, :, by + J:end:S, bx + J:end:S] for i, (bx, by) in enumerate(cells)], dim=0)
    assert torch.equal(g, ref)
    for mixture in (False, True):
        outs = torch.randn((2 if mixture else 1) * S * S * N, C, h0, h0)
        xglob = torch.zeros(N, C, Hp, Wp)
        for i, (bx, by) in enumerate(cells + cells if mixture else cells):
            xglob[:, :, by + J:end:S, bx + J:end:S] += outs[i * N:(i + 1) * N]
        c2 = torch.tensor(0.37) ** 1.0
        want = want * 0 + (buf / torch.where(cnt == 0, torch.tensor(1), cnt)) * (1 - c2) + ((xglob / 2 if mixture else xglob) / 1) * c2
        got = E.demofusion_combine(E.window_blend(tiles.to(cuda), ws, N, C, Hp, Wp), outs.to(cuda), S, J, mixture, float(c2)).cpu()
        assert torch.equal(got, want)
