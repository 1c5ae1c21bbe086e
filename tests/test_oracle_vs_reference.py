"""Pins the oracle: runs the UPSTREAM reference code itself (imported verbatim from /root/reference under
hostsim/stub_host.py) next to the restatement on the same seeded inputs.  Skipped where /root/reference is not mounted
(the GPU box) -- there the committed golden vectors (test_oracle_golden.py) carry the pin."""
import pytest
import torch

from oracle import blend_oracle as bo
from hostsim import ldm_decoder as ld
from hostsim import stub_host as sh
from oracle import vae_oracle as vo

pytestmark = pytest.mark.skipif(not sh.reference_available(), reason="/root/reference not mounted")

import os, sys  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as mg  # noqa: E402  (shares the reference driver with the fixture generator)


@pytest.fixture(scope="module")
def ref():
    return sh.load_reference()


@pytest.mark.parametrize("tw,th", [(96, 96), (128, 64), (33, 57), (16, 24)])
def test_gaussian_weights(ref, tw, th):
    assert torch.equal(ref.utils.gaussian_weights(tw, th), bo.gaussian_weights(tw, th))


@pytest.mark.parametrize("w,h,r", [(40, 30, 0.2), (41, 33, 0.5), (10, 10, 0.0), (64, 64, 1.0), (7, 9, 0.9), (30, 52, 0.7)])
def test_feather_mask(ref, w, h, r):
    assert torch.equal(ref.utils.feather_mask(w, h, r), bo.feather_mask(w, h, r))


@pytest.mark.parametrize("args", mg.GRID_CASES)
def test_grid(ref, args):
    w, h, tw, th, ov, bs = args
    d = ref.multidiffusion.MultiDiffusion(sh.make_processing(w * 8, h * 8), sh.kdiff_sampler())
    d.init_grid_bbox(tw, th, ov, bs)
    boxes, batches, tw2, th2, ov2 = bo.init_grid(w, h, tw, th, ov, bs)
    assert [[b.x, b.y, b.w, b.h] for batch in d.batched_bboxes for b in batch] == [list(b) for b in boxes]
    assert [len(b) for b in d.batched_bboxes] == [len(b) for b in batches]
    assert torch.equal(d.weights, bo.grid_weight_map(w, h, boxes, 1.0))


@pytest.mark.parametrize("case", mg.BLEND_CASES, ids=lambda c: c["name"])
def test_blend_bit_exact(ref, case):
    out, weights = mg.run_ref_blend(ref, case)
    regs = [bo.Region(*bo.region_rect(case["W"], case["H"], fx, fy, fw, fh), mode, fr)
            for (fx, fy, fw, fh, mode, fr) in (case["regions"] or [])]
    o = bo.BlendOracle(case["method"], case["W"], case["H"], case["tw"], case["th"], case["ov"], case["bs"], regs, case["bg"])
    torch.manual_seed(case["seed"])
    x = torch.randn(case["N"], 4, case["H"], case["W"])
    mine = o.evaluate(x, bo.synthetic_denoiser, bo.synthetic_region_denoiser)
    assert torch.equal(weights, o.weights)
    assert torch.equal(out, mine)


@pytest.mark.parametrize("args", mg.TILE_CASES)
def test_split_tiles(ref, args):
    h, w, ts, is_dec = args
    hook = ref.tilevae.VAEHook(None, ts, is_decoder=is_dec, fast_decoder=True, fast_encoder=True, color_fix=False)
    assert hook.split_tiles(h, w) == vo.split_tiles(h, w, ts, is_dec)


def test_task_queue_shape(ref):
    dec = ld.make_decoder(0, small=True)
    q = ref.tilevae.build_task_queue(dec, True)
    ops = vo.build_ops(dec)
    assert len(q) == len(ops) == 123
    assert sum(1 for t in q if t[0] == "pre_norm") == sum(1 for k, _ in ops if k == "norm") == 30


@pytest.mark.parametrize("H,W,ts,fast", [(40, 56, 16, True), (40, 56, 16, False), (30, 70, 24, True), (60, 34, 16, False),
                                         (48, 48, 32, True)])
def test_tiled_decode_bit_exact(ref, H, W, ts, fast):
    dec = ld.make_decoder(0, small=True)
    dec.original_forward = dec.forward
    torch.manual_seed(2)
    z = torch.randn(1, 4, H, W)
    hook = ref.tilevae.VAEHook(dec, ts, is_decoder=True, fast_decoder=fast, fast_encoder=False, color_fix=False)
    assert torch.equal(hook(z), vo.tiled_forward(dec, z, ts, fast))


@pytest.mark.parametrize("H,W,ts,fast,color_fix", [(160, 192, 64, True, False), (160, 192, 64, False, False),
                                                   (136, 200, 64, True, True), (200, 120, 96, True, False)])
def test_tiled_encode_bit_exact(ref, H, W, ts, fast, color_fix):
    """Encoder direction (pad 32, stride-2 Downsample, color_fix semi-fast mode) -- upstream tilevae.py:155-171, 492-496."""
    enc = ld.make_encoder(0, small=True)
    enc.original_forward = enc.forward
    torch.manual_seed(4)
    x = torch.randn(1, 3, H, W)
    hook = ref.tilevae.VAEHook(enc, ts, is_decoder=False, fast_decoder=False, fast_encoder=fast, color_fix=color_fix)
    assert torch.equal(hook(x), vo.tiled_forward(enc, x, ts, fast, is_decoder=False, color_fix=color_fix))


def test_encoder_task_queue_shape(ref):
    enc = ld.make_encoder(0, small=True)
    q = ref.tilevae.build_task_queue(enc, False)
    ops = vo.build_ops(enc, False)
    assert len(q) == len(ops)
    assert sum(1 for t in q if t[0] == "pre_norm") == sum(1 for k, _ in ops if k == "norm")


REGION_NOISE = [  # fx, fy, fw, fh, mode, seed   (overlapping backgrounds AND overlapping foregrounds, partly off-canvas)
    (0.0, 0.0, 0.5, 1.0, "Background", 11), (0.3, 0.0, 0.5, 1.0, "Background", 12), (0.6, 0.1, 0.6, 0.8, "Foreground", 13),
    (0.55, 0.3, 0.2, 0.5, "Foreground", 14), (0.1, 0.7, 0.25, 0.2, "Background", 15),
]


def test_region_noise_bit_exact(ref):
    """create_random_tensors_hijack (upstream tilediffusion.py:486-529) == oracle.region_noise."""
    td = ref.tilediffusion
    if td is None:
        pytest.skip("upstream scripts/tilediffusion.py does not import under the stub host")
    U = ref.utils
    torch.manual_seed(5)
    org = torch.randn(2, 4, 40, 56)
    td.Script.create_random_tensors_original_md = staticmethod(lambda *a, **k: org.clone())
    try:
        settings = {i: U.BBoxSettings(True, fx, fy, fw, fh, "", "", mode, 0.2, seed) for i, (fx, fy, fw, fh, mode, seed) in enumerate(REGION_NOISE)}
        info = {f"Region {i + 1}": {} for i in settings}
        got = td.Script().create_random_tensors_hijack(settings, info, (4, 40, 56), [1, 2])
    finally:
        del td.Script.create_random_tensors_original_md
    assert torch.equal(got, bo.region_noise(org, REGION_NOISE))
    assert [info[f"Region {i + 1}"]["seed"] for i in range(len(REGION_NOISE))] == [r[5] for r in REGION_NOISE]


def test_gn_and_attn_primitives(ref):
    torch.manual_seed(11)
    t = torch.randn(2, 64, 9, 13) * 3 + 0.5
    v1, m1 = ref.tilevae.get_var_mean(t, 32)
    v2, m2 = vo.get_var_mean(t, 32)
    assert torch.equal(v1, v2) and torch.equal(m1, m2)
    g, b = torch.randn(64), torch.randn(64)
    assert torch.equal(ref.tilevae.custom_group_norm(t, 32, m1, v1, g, b), vo.custom_group_norm(t, 32, m2, v2, g, b))
    torch.manual_seed(12)
    ab = ld.AttnBlock(64).eval()
    hx = torch.randn(1, 64, 7, 9)
    with torch.no_grad():
        assert torch.equal(ref.attn.attn_forward(ab, hx), vo.attn_body(ab, hx))


# ---------------------------------------------------------------------------------------------------------------------
# Noise Inversion: renoise composite of upstream's sample_img2img (abstractdiffusion.py:606-681) -- the upstream method itself is
# run under stubs for everything around the composite (cached inversion latent, sigma schedule, retouch mask) and the noise it
# hands to the original sample_img2img is compared with oracle.noise_inverse_blend.
# ---------------------------------------------------------------------------------------------------------------------
NI_REGIONS = [  # x, y, w, h (latent px), mode, feather
    (3, 2, 20, 12, "Background", 0.2), (10, 6, 18, 14, "Foreground", 0.3), (16, 10, 20, 12, "Foreground", 0.6), (0, 14, 9, 10, "Background", 0.2),
]


def _ni_inputs(W, H, seed=7):
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(2, 4, H, W, generator=g)
    init_latent = torch.randn(2, 4, H, W, generator=g)
    xt = torch.randn(2, 4, H, W, generator=g) * 3.0
    np_mask = torch.rand(H * 8, W * 8, generator=g).numpy()
    sigmas = torch.linspace(7.5, 0.03, 9)
    return noise, init_latent, xt, np_mask, sigmas


def _ni_mask(np_mask, H, W, strength):
    import torch.nn.functional as F
    m = 1 - F.interpolate(torch.from_numpy(np_mask).unsqueeze(0).unsqueeze(0), size=(H, W), mode="bilinear").squeeze(0).squeeze(0)
    m *= strength
    return torch.clamp(m, 0, 1)


@pytest.mark.parametrize("grid,strength", [(True, 0.7), (False, 0.7), (False, 1.6), (True, 0.0)])
def test_noise_inverse_composite_bit_exact(ref, grid, strength):
    from types import SimpleNamespace
    from PIL import Image
    absd, utils = ref.abstractdiffusion, ref.utils
    W, H = 40, 28
    noise, init_latent, xt, np_mask, sigmas = _ni_inputs(W, H)
    p = sh.make_processing(W * 8, H * 8)
    p.init_images = [Image.new("RGB", (W * 8, H * 8))]
    p.sd_model = SimpleNamespace(sd_model_hash="hash")
    p.init_latent = init_latent
    d = ref.multidiffusion.MultiDiffusion(p, sh.kdiff_sampler())
    if grid:
        d.init_grid_bbox(16, 16, 4, 2)
    d.enable_grid_bbox = grid
    d.custom_bboxes = [utils.CustomBBox(x, y, w, h, "", "", mode, fr, 1) for (x, y, w, h, mode, fr) in NI_REGIONS]
    d.noise_inverse_steps, d.noise_inverse_retouch = 5, 1.0
    d.noise_inverse_renoise_strength, d.noise_inverse_renoise_kernel = strength, 3
    d.noise_inverse_get_cache = lambda: utils.NoiseInverseCache("hash", init_latent.clone(), xt, 5, 1.0, [""])
    captured = {}
    d.sample_img2img_original = lambda p_, x_, n_, c_, uc_, steps_, ic_: captured.setdefault("noise", n_)
    old = (getattr(absd, "get_retouch_mask", None), getattr(absd.sd_samplers_common, "setup_img2img_steps", None))
    absd.get_retouch_mask = lambda img, k: np_mask
    absd.sd_samplers_common.setup_img2img_steps = lambda p_, steps: (steps or 8, 6)
    try:
        sampler = SimpleNamespace(get_sigmas=lambda p_, steps: sigmas)
        d.sample_img2img(sampler, p, torch.zeros_like(noise), noise, None, None, steps=8, image_conditioning=None)
    finally:
        absd.get_retouch_mask = old[0]
        if old[1] is None:
            del absd.sd_samplers_common.setup_img2img_steps
        else:
            absd.sd_samplers_common.setup_img2img_steps = old[1]
    inverse = xt - init_latent / sigmas[0]
    regs = [bo.Region(x, y, w, h, mode, fr) for (x, y, w, h, mode, fr) in NI_REGIONS]
    mine = bo.noise_inverse_blend(noise, inverse, _ni_mask(np_mask, H, W, strength) if strength > 0 else None, regs, grid)
    assert torch.equal(captured["noise"], mine)


# ---------------------------------------------------------------------------------------------------------------------
# Region forwards: the product's kdiff_custom_forward / ddim_custom_forward against upstream's under every batching mode of the
# host's CFG denoiser (whole batch, prompts of different token length, partial batches, edit model).  Pure host logic: runs
# without the engine.  The "model" records (rows, prompt rows) per call and returns a function of both, so routing errors show.
# ---------------------------------------------------------------------------------------------------------------------
def _fake_forward(calls):
    def f(x, sigma, cond):
        t = cond["c_crossattn"][0] if isinstance(cond["c_crossattn"], list) else cond["c_crossattn"]
        assert t.shape[0] == x.shape[0] == sigma.shape[0], f"batch mismatch: x {x.shape[0]} cond {t.shape[0]} sigma {sigma.shape[0]}"
        calls.append((x.shape[0], t.shape[1]))
        return x * 0.5 + t.mean(dim=(1, 2)).view(-1, 1, 1, 1) + sigma.view(-1, 1, 1, 1)
    return f


def _drive_custom_forward(mods, lens, chunks, batch_cond_uncond, edit, glob=(77, 77)):
    """mods: namespace with multidiffusion / utils (reference or product).  lens = (region prompt tokens, region negative tokens);
    chunks = row counts of the successive model calls of one sampler step."""
    from types import SimpleNamespace
    _, shared = sh.host()
    old_bcu = getattr(shared, "batch_cond_uncond", True)
    shared.batch_cond_uncond = batch_cond_uncond
    cls = mods.multidiffusion.MultiDiffusion
    _missing = object()
    old_edit = cls.__dict__.get("is_edit_model", _missing)
    cls.is_edit_model = edit
    C = mods.utils.Condition
    old_rc, old_ru = C.reconstruct_cond, C.reconstruct_uncond
    g = torch.Generator().manual_seed(5)
    tens = {"gc": torch.randn(1, glob[0], 8, generator=g), "gu": torch.randn(1, glob[1], 8, generator=g),
            "rc": torch.randn(1, lens[0], 8, generator=g), "ru": torch.randn(1, lens[1], 8, generator=g)}
    C.reconstruct_cond = staticmethod(lambda c, step: tens[c])
    C.reconstruct_uncond = staticmethod(lambda c, step: tens[c])
    try:
        p = sh.make_processing(64 * 8, 48 * 8)
        smp = sh.kdiff_sampler()
        smp.model_wrap_cfg = SimpleNamespace(step=3, inner_model=SimpleNamespace(forward=None), image_cfg_scale=None)
        d = cls(p, smp)
        d.custom_bboxes = [mods.utils.CustomBBox(4, 4, 16, 12, "", "", "Background", 0.2, 1)]
        d.custom_bboxes[0].cond, d.custom_bboxes[0].uncond = "rc", "ru"
        d.cond_basis, d.uncond_basis = "gc", "gu"
        calls, outs = [], []
        total = sum(chunks)
        x = torch.randn(total, 4, 12, 16, generator=g)
        sig = torch.rand(total, generator=g)
        cond = {"c_crossattn": [torch.zeros(total, 77, 8)], "c_concat": [torch.zeros(total, 5, 1, 1)]}
        a = 0
        for n in chunks:
            sub = {"c_crossattn": [cond["c_crossattn"][0][a:a + n]], "c_concat": [cond["c_concat"][0][a:a + n]]}
            outs.append(d.kdiff_custom_forward(x[a:a + n], sig[a:a + n], sub, 0, d.custom_bboxes[0], _fake_forward(calls)))
            a += n
        return torch.cat(outs), calls
    finally:
        C.reconstruct_cond, C.reconstruct_uncond = old_rc, old_ru
        if old_edit is _missing:
            del cls.is_edit_model
        else:
            cls.is_edit_model = old_edit
        shared.batch_cond_uncond = old_bcu


KDIFF_CASES = [  # (region prompt tokens, region negative tokens), chunks, batch_cond_uncond, edit, global token lengths
    ((77, 77), [2], True, False, (77, 77)),        # whole batch, equal lengths: one call
    ((154, 77), [2], True, False, (77, 77)),       # region prompt > 75 tokens: two calls
    ((77, 77), [3], True, True, (77, 77)),         # edit model: [cond, uncond, uncond]
    ((154, 77), [3], True, True, (77, 77)),        # edit model + different lengths
    ((77, 77), [1, 1], False, False, (77, 77)),    # batch_cond_uncond off: cond, then uncond
    ((154, 77), [1, 1], False, False, (77, 77)),
    ((77, 77), [1, 1], True, False, (154, 77)),    # GLOBAL prompts of different length: the host itself splits
]


@pytest.mark.parametrize("lens,chunks,bcu,edit,glob", KDIFF_CASES)
def test_kdiff_custom_forward_matches_upstream(ref, lens, chunks, bcu, edit, glob):
    want, calls_ref = _drive_custom_forward(ref, lens, chunks, bcu, edit, glob)
    got, calls = _drive_custom_forward(sh.load_plugin(), lens, chunks, bcu, edit, glob)
    assert calls == calls_ref, f"model calls (rows, tokens): product {calls} vs upstream {calls_ref}"
    assert torch.equal(got, want)


@pytest.mark.parametrize("lens", [(77, 77), (154, 77), (77, 154)])
def test_ddim_custom_forward_matches_upstream(ref, lens):
    from types import SimpleNamespace
    outs = []
    for mods in (ref, sh.load_plugin()):
        C = mods.utils.Condition
        old_rc, old_ru = C.reconstruct_cond, C.reconstruct_uncond
        g = torch.Generator().manual_seed(9)
        tens = {"rc": torch.randn(1, lens[0], 8, generator=g), "ru": torch.randn(1, lens[1], 8, generator=g)}
        C.reconstruct_cond = staticmethod(lambda c, step: tens[c])
        C.reconstruct_uncond = staticmethod(lambda c, step: tens[c])
        try:
            smp = sh.kdiff_sampler()
            smp.model_wrap_cfg = SimpleNamespace(step=3, inner_model=SimpleNamespace(forward=None), image_cfg_scale=None)
            d = mods.multidiffusion.MultiDiffusion(sh.make_processing(512, 384), smp)
            bbox = mods.utils.CustomBBox(4, 4, 16, 12, "", "", "Background", 0.2, 1)
            bbox.cond, bbox.uncond = "rc", "ru"
            seen = {}

            def fwd(x, c, ts, unconditional_conditioning):
                seen["c"], seen["uc"] = c, unconditional_conditioning
                return x
            cond_in = {"c_crossattn": [torch.zeros(1, 77, 8)], "c_concat": [torch.zeros(1, 5, 1, 1)]}
            d.ddim_custom_forward(torch.zeros(1, 4, 12, 16), cond_in, bbox, torch.zeros(1), fwd)
            outs.append((seen["c"]["c_crossattn"][0], seen["uc"]["c_crossattn"][0]))
        finally:
            C.reconstruct_cond, C.reconstruct_uncond = old_rc, old_ru
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[1][0].shape[1] == outs[1][1].shape[1]


# ---------------------------------------------------------------------------------------------------------------------
# DemoFusion: upstream's delegate (views with jitter, sample_one_step) against oracle/demofusion_oracle.py
# ---------------------------------------------------------------------------------------------------------------------
def _demo_tile_fn(x):
    return 0.9 * x + 0.1 * x.flip(-1) + 0.05 * x.flip(-2)


DEMO_CASES = [  # W0, H0 (latent size of the base image), scale S, window, overlap, jitter, mixture, gaussian
    (24, 24, 2, 16, 8, True, False, True),
    (24, 24, 3, 16, 8, True, True, True),
    (24, 24, 2, 16, 8, False, False, True),
    (20, 20, 2, 16, 4, True, True, True),
]


@pytest.mark.parametrize("W0,H0,S,window,overlap,jitter,mixture,gaussian", DEMO_CASES)
def test_demofusion_sample_one_step_bit_exact(ref, W0, H0, S, window, overlap, jitter, mixture, gaussian):
    import random
    from types import SimpleNamespace
    from oracle import demofusion_oracle as do
    if ref.demofusion is None:
        pytest.skip("upstream demofusion not importable")
    W, H = W0 * S, H0 * S
    p = sh.make_processing(W * 8, H * 8)
    p.random_jitter, p.mixture, p.current_scale_num, p.gaussian_filter = jitter, mixture, S, gaussian
    p.cosine_scale_2, p.cosine_scale_3 = 1.0, 1.0
    p.sd_model = SimpleNamespace(apply_model=lambda x, t, cond: _demo_tile_fn(x))
    smp = sh.kdiff_sampler()
    smp.model_wrap_cfg = SimpleNamespace(step=0, inner_model=SimpleNamespace(forward=None), image_cfg_scale=None, forward=None)
    d = ref.demofusion.DemoFusion(p, smp)
    d.window_size = window
    d.sig = 0.3
    d.w, d.h = W, H
    random.seed(1234)
    d.get_views(overlap, 3, 2)
    d.sampler_forward = lambda x, sigma, cond: _demo_tile_fn(x)
    d.repeat_3 = False
    d.cosine_factor = 0.5 * (1 + torch.cos(torch.pi * torch.tensor((3 + 1) / (10 + 1))))
    J = d.jitter_range
    torch.manual_seed(3)
    x = torch.randn(2, 4, H + 2 * J, W + 2 * J)
    cond = {"c_crossattn": [torch.zeros(2, 77, 8)], "c_concat": [torch.zeros(2, 5, 1, 1)]}
    want = d.sample_one_step(x.clone(), torch.ones(2), cond)
    random.seed(1234)
    origins, J2, ov2, stride = do.views(W, H, window, overlap, jitter)
    assert J2 == J and [(b.x, b.y) for bb in d.batched_bboxes for b in bb] == origins
    got = do.sample_one_step(x.clone(), origins, window, J, 3, 2, S, mixture, gaussian, 0.3, d.cosine_factor, 1.0, 1.0, _demo_tile_fn)
    assert torch.equal(got, want)
