"""Pins the oracle: runs the UPSTREAM reference code itself (imported verbatim from /root/reference under
oracle/stub_host.py) next to the restatement on the same seeded inputs.  Skipped where /root/reference is not mounted
(the GPU box) -- there the committed golden vectors (test_oracle_golden.py) carry the pin."""
import pytest
import torch

from oracle import blend_oracle as bo
from oracle import ldm_decoder as ld
from oracle import stub_host as sh
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
