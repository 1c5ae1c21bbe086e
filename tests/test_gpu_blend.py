"""GPU parity of the tile gather / weight maps / overlap blend (HIP kernels through the C ABI and the plugin delegates)
against the oracle and the upstream-generated golden vectors.  fp32 results are required to be BIT-EXACT: the blend kernel
sums in upstream's tile order with unfused multiply/add, so there is nothing to tolerate."""
import numpy as np
import pytest
import torch

from oracle import blend_oracle as bo
from hostsim import stub_host as sh

pytestmark = pytest.mark.gpu


def _delegate(plugin, case_or_method, W, H, tw, th, ov, bs, regions=None, bg=True):
    method = case_or_method
    cls = plugin.multidiffusion.MultiDiffusion if method == "md" else plugin.mixtureofdiffusers.MixtureOfDiffusers
    d = cls(sh.make_processing(W * 8, H * 8), sh.kdiff_sampler())
    if bg or not regions:
        d.init_grid_bbox(tw, th, ov, bs)
    if regions:
        U = plugin.utils
        settings = {i: U.BBoxSettings(True, fx, fy, fw, fh, "", "", mode, fr, -1) for i, (fx, fy, fw, fh, mode, fr) in enumerate(regions)}
        d.init_custom_bbox(settings, bg, False)
    d.init_done()
    if d.pbar is not None:
        d.pbar.close()
    d.update_pbar = lambda: None
    return d


def _evaluate(plugin, d, method, x):
    dev = x.device
    if method == "md":
        return d.sample_one_step(x, None, lambda xt, b: bo.synthetic_denoiser(xt), lambda xr, i, b: bo.synthetic_region_denoiser(xr, i))
    _, shared = sh.host()
    shared.sd_model.apply_model_original_md = lambda x_, t_, c_: bo.synthetic_denoiser(x_)
    d.custom_apply_model = lambda x_in, t_in, c_in, bbox_id, bbox: bo.synthetic_region_denoiser(x_in, bbox_id)
    N = x.shape[0]
    cond = {"c_crossattn": [torch.zeros(N, 77, 768, device=dev)], "c_concat": [torch.zeros(N, 5, 1, 1, device=dev)]}
    return d.apply_model_hijack(x, torch.zeros(N, device=dev), cond)


def test_maps_match_upstream_goldens(plugin, cuda, golden_maps):
    U = plugin.utils
    for key in golden_maps.files:
        kind, dims = key.split("_", 1)
        if kind == "gauss":
            tw, th = map(int, dims.split("x"))
            got = U.gaussian_weights(tw, th)
        else:
            wh, r = dims.rsplit("_", 1)
            w, h = map(int, wh.split("x"))
            got = U.feather_mask(w, h, float(r))
        assert got.is_cuda and got.dtype == torch.float32
        assert np.array_equal(got.cpu().numpy(), golden_maps[key]), key


@pytest.mark.parametrize("args", [(256, 256, 96, 96, 48), (100, 70, 32, 24, 8), (97, 131, 16, 16, 12), (512, 128, 96, 96, 48)])
def test_weight_maps(plugin, cuda, args):
    w, h, tw, th, ov = args
    E = plugin.engine
    plan = E.Plan(w, h, tw, th, ov, 4)
    boxes = [tuple(b) for b in plan.bboxes]
    for tile_w in (None, bo.gaussian_weights(plan.tile_w, plan.tile_h)):
        wm = torch.zeros(1, 1, h, w, device=cuda)
        E.weight_map_add_grid(plan, None if tile_w is None else tile_w.to(cuda), wm)
        ref = bo.grid_weight_map(w, h, boxes, 1.0 if tile_w is None else tile_w)
        assert torch.equal(wm.cpu(), ref)
        rc = E.reciprocal(wm)
        assert torch.equal(rc.cpu(), 1 / ref)
    # split_bboxes of the plugin surface (raw grid)
    bl, wm2 = plugin.utils.split_bboxes(w, h, plan.tile_w, plan.tile_h, plan.overlap, 1.0)
    assert [(b.x, b.y, b.w, b.h) for b in bl] == boxes and torch.equal(wm2.cpu(), bo.grid_weight_map(w, h, boxes, 1.0))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_gather(plugin, cuda, dtype):
    E = plugin.engine
    torch.manual_seed(0)
    x = torch.randn(3, 4, 70, 100).to(dtype)
    plan = E.Plan(100, 70, 32, 24, 8, 3)
    xs = x.to(cuda)
    tiles = E.gather_all(plan, xs)
    assert len(tiles) == plan.num_batches
    for b, batch in enumerate(plan.batches):
        ref = torch.cat([x[:, :, by:by + bh, bx:bx + bw] for (bx, by, bw, bh) in batch], dim=0)
        assert torch.equal(tiles[b].cpu(), ref)
        assert torch.equal(E.gather(plan, xs, b).cpu(), ref)
    packed = torch.empty(plan.num_tiles * 3, 4, plan.tile_h, plan.tile_w, dtype=dtype, device=cuda)
    E.gather_all(plan, xs, [packed])
    assert torch.equal(packed.cpu(), torch.cat([t.cpu() for t in tiles], dim=0))
    assert torch.equal(E.gather_rect(xs, 7, 5, 33, 21).cpu(), x[:, :, 5:26, 7:40])


def test_blend_matches_upstream_goldens_bit_exact(plugin, cuda, cases, golden_blend):
    for c in cases["blend"]:
        d = _delegate(plugin, c["method"], c["W"], c["H"], c["tw"], c["th"], c["ov"], c["bs"], c["regions"], c["bg"])
        torch.manual_seed(c["seed"])
        x = torch.randn(c["N"], 4, c["H"], c["W"])
        out = _evaluate(plugin, d, c["method"], x.to(cuda))
        assert np.array_equal(d.weights.cpu().numpy(), golden_blend[c["name"] + "/weights"]), c["name"]
        got, ref = out.cpu().numpy(), golden_blend[c["name"] + "/out"]
        assert np.array_equal(got, ref), f"{c['name']}: max abs diff {np.abs(got - ref).max()}"


CONFIGS = [  # BASELINE.json configs at full latent size
    ("cfg2", "md", 256, 256, 96, 96, 48, 4), ("cfg3", "mod", 512, 512, 96, 96, 48, 4), ("cfg3b", "mod", 512, 512, 96, 96, 8, 4),
    ("cfg4_md", "md", 1024, 1024, 128, 128, 8, 4), ("cfg4_mod", "mod", 1024, 1024, 128, 128, 8, 4),
    ("cfg4_ov64", "mod", 1024, 1024, 128, 128, 64, 4), ("cfg4_ov64_md", "md", 1024, 1024, 128, 128, 64, 4),
    # 441 tile batches: more than the 320 pointers that ride in the kernel arguments -> the packed-buffer form behind the same calls
    ("b441_md", "md", 1024, 1024, 96, 96, 48, 1), ("b441_mod", "mod", 1024, 1024, 96, 96, 48, 1),
    # rectangular canvases and tiles; regular grids (every origin = index x stride: 48 x 80, 56 x 56 -- cfg4 above is one too) and one that is
    # regular along y only
    ("reg_rect_md", "md", 640, 416, 64, 96, 16, 4), ("reg_rect_mod", "mod", 640, 416, 64, 96, 16, 4), ("reg_sq_md", "md", 512, 512, 64, 64, 8, 4),
    ("reg_y_only_mod", "mod", 512, 416, 96, 96, 16, 4),
]


@pytest.mark.parametrize("name,method,W,H,tw,th,ov,bs", CONFIGS)
def test_blend_full_size_vs_oracle(plugin, cuda, name, method, W, H, tw, th, ov, bs):
    d = _delegate(plugin, method, W, H, tw, th, ov, bs)
    o = bo.BlendOracle(method, W, H, tw, th, ov, bs)
    torch.manual_seed(0)
    x = torch.randn(2, 4, H, W)
    out = _evaluate(plugin, d, method, x.to(cuda)).cpu()
    ref = o.evaluate(x, bo.synthetic_denoiser)
    assert torch.equal(d.weights.cpu(), o.weights)
    assert torch.equal(out, ref), f"{name}: max abs diff {(out - ref).abs().max().item()}"
    # size-independent property: an identity denoiser must give the input back (partition of unity)
    if method == "md":
        ident = d.sample_one_step(x.to(cuda), None, lambda xt, b: xt, None).cpu()
    else:
        _, shared = sh.host()
        shared.sd_model.apply_model_original_md = lambda x_, t_, c_: x_
        cond = {"c_crossattn": [torch.zeros(2, 77, 768, device=cuda)], "c_concat": [torch.zeros(2, 5, 1, 1, device=cuda)]}
        ident = d.apply_model_hijack(x.to(cuda), torch.zeros(2, device=cuda), cond).cpu()
    assert (ident - x).abs().max().item() < 2e-6


def _random_geometry(seed):
    """Canvas, tile, overlap, tile batch, method and image batch drawn from what the delegates accept (upstream's sliders: tile 16..256 in
    steps of 16 by the UI, any integer through the API; overlap 0..tile - 4): odd canvases, tiles larger than the canvas (clamped like
    upstream's split_bboxes), ragged last rows / columns, origins of every alignment."""
    import random
    r = random.Random(seed)
    W, H = r.randint(24, 520), r.randint(24, 520)
    tw, th = r.choice([16, 24, 40, 64, 72, 96, 100, 128, 160]), r.choice([16, 24, 40, 64, 72, 96, 100, 128, 160])
    ov = r.randint(0, max(0, min(tw, th, W, H) - 4))
    return ("md", "mod")[seed % 2], W, H, tw, th, ov, r.randint(1, 8), r.randint(1, 3)


@pytest.mark.parametrize("seed", range(32))
def test_blend_random_geometries_vs_oracle(plugin, cuda, seed):
    """Bit-exact against the oracle on 32 seeded random geometries: which of the blend kernels runs (k_blend's vector / element path,
    k_blend_lds on grids with odd origins, the packed-pointer form) follows from the geometry, the sum must not."""
    method, W, H, tw, th, ov, bs, N = _random_geometry(seed)
    d = _delegate(plugin, method, W, H, tw, th, ov, bs)
    o = bo.BlendOracle(method, W, H, tw, th, ov, bs)
    torch.manual_seed(seed)
    x = torch.randn(N, 4, H, W)
    out = _evaluate(plugin, d, method, x.to(cuda)).cpu()
    ref = o.evaluate(x, bo.synthetic_denoiser)
    assert torch.equal(d.weights.cpu(), o.weights), (method, W, H, tw, th, ov, bs, N)
    # (Mixture of Diffusers with small tiles: upstream's Gaussian underflows to 0 on the outermost rows, its 1 / weights is inf there and
    # the blended pixel NaN -- upstream behaviour, reproduced: NaNs must sit at the same pixels, everything else must be the same bits)
    nan = torch.isnan(ref)
    assert torch.equal(torch.isnan(out), nan), f"{(method, W, H, tw, th, ov, bs, N)}: NaN pattern differs"
    z = torch.zeros(())
    assert torch.equal(torch.where(nan, z, out), torch.where(nan, z, ref)), \
        f"{(method, W, H, tw, th, ov, bs, N)}: max abs diff {(torch.where(nan, z, out) - torch.where(nan, z, ref)).abs().max().item()}"


def test_blend_cfg5_regions_vs_oracle(plugin, cuda):
    regs = [(0.0, 0.0, 0.4, 1.0, "Background", 0.2), (0.3, 0.0, 0.4, 1.0, "Background", 0.2), (0.6, 0.1, 0.4, 0.8, "Foreground", 0.2)]
    W, H = 512, 128
    for method in ("md", "mod"):
        d = _delegate(plugin, method, W, H, 96, 96, 48, 4, regs, True)
        oregs = [bo.Region(*bo.region_rect(W, H, fx, fy, fw, fh), mode, fr) for (fx, fy, fw, fh, mode, fr) in regs]
        o = bo.BlendOracle(method, W, H, 96, 96, 48, 4, oregs, True)
        torch.manual_seed(3)
        x = torch.randn(2, 4, H, W)
        out = _evaluate(plugin, d, method, x.to(cuda)).cpu()
        ref = o.evaluate(x, bo.synthetic_denoiser, bo.synthetic_region_denoiser)
        assert torch.equal(d.weights.cpu(), o.weights)
        assert torch.equal(out, ref), f"{method}: max abs diff {(out - ref).abs().max().item()}"


def test_blend_packed_partial_and_row_ranges(plugin, cuda):
    """Engine-level modes used by the multi-GPU path: packed tile buffer, tile-range partial sums + finalize, row bands."""
    E = plugin.engine
    W, H, tw, th, ov, bs = 160, 120, 48, 40, 16, 4
    for method, code in (("md", E.METHOD_MD), ("mod", E.METHOD_MOD)):
        o = bo.BlendOracle(method, W, H, tw, th, ov, bs)
        plan = E.Plan(W, H, tw, th, ov, bs)
        torch.manual_seed(1)
        x = torch.randn(2, 4, H, W)
        ref = o.evaluate(x, bo.synthetic_denoiser)
        outs = [bo.synthetic_denoiser(o.gather(x, b)).to(cuda) for b in o.batches]
        kw = dict(weights=o.weights.to(cuda)) if method == "md" else dict(tile_w=o.tile_weights.to(cuda), rescale=o.rescale.to(cuda))
        full = E.blend(plan, code, outs, 2, 4, **kw)
        assert torch.equal(full.cpu(), ref)
        packed = torch.cat(outs, dim=0).contiguous()
        assert torch.equal(E.blend(plan, code, [packed], 2, 4, packed=True, **kw).cpu(), ref)
        # two "ranks": tile rows split in two bands -> partial sums -> add -> finalize
        half = (plan.rows // 2) * plan.cols
        p0 = E.blend(plan, code, outs, 2, 4, partial=True, tile_range=(0, half), **kw)
        p1 = E.blend(plan, code, outs, 2, 4, partial=True, tile_range=(half, plan.num_tiles), **kw)
        fin = E.blend_finalize(plan, code, (p0 + p1).contiguous(), weights=kw.get("weights"))
        assert torch.allclose(fin.cpu(), ref, rtol=1e-6, atol=1e-6)
        # row band only
        band = torch.full((2, 4, H, W), float("nan"), device=cuda)
        E.blend(plan, code, outs, 2, 4, out=band, row_range=(30, 77), **kw)
        assert torch.equal(band[:, :, 30:77].cpu(), ref[:, :, 30:77]) and torch.isnan(band[:, :, :30]).all() and torch.isnan(band[:, :, 77:]).all()


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
def test_blend_half_io_accumulates_in_fp32(plugin, cuda, dtype, tol):
    W, H = 128, 96
    for method in ("md", "mod"):
        d = _delegate(plugin, method, W, H, 48, 48, 24, 4)
        o = bo.BlendOracle(method, W, H, 48, 48, 24, 4)
        torch.manual_seed(5)
        x = torch.randn(2, 4, H, W).to(dtype)
        out = _evaluate(plugin, d, method, x.to(cuda))
        assert out.dtype == dtype
        ref = o.evaluate(x.float(), lambda t: bo.synthetic_denoiser(t.to(dtype).float()).to(dtype).float())
        err = (out.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err < tol, f"{method} {dtype}: rel err {err}"


def test_errors_are_reported_not_crashed(plugin, cuda):
    E = plugin.engine
    plan = E.Plan(64, 64, 32, 32, 8, 4)
    with pytest.raises(E.MdtileError, match="batches given"):
        E.blend(plan, E.METHOD_MD, [torch.zeros(8, 4, 32, 32, device=cuda)], 2, 4, weights=torch.ones(64 * 64, device=cuda))
    with pytest.raises(E.MdtileError, match="needs d_weights"):
        E.blend(plan, E.METHOD_MD, [torch.zeros(8, 4, 32, 32, device=cuda)] * plan.num_batches, 2, 4)
    with pytest.raises(E.MdtileError, match="outside"):
        E.gather_rect(torch.zeros(1, 4, 16, 16, device=cuda), 10, 10, 8, 8)


REGION_NOISE = [  # fx, fy, fw, fh, mode, seed   (overlapping backgrounds AND overlapping foregrounds, partly off-canvas)
    (0.0, 0.0, 0.5, 1.0, "Background", 11), (0.3, 0.0, 0.5, 1.0, "Background", 12), (0.6, 0.1, 0.6, 0.8, "Foreground", 13),
    (0.55, 0.3, 0.2, 0.5, "Foreground", 14), (0.1, 0.7, 0.25, 0.2, "Background", 15),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_region_noise_hijack_bit_exact(plugin, cuda, dtype):
    """Script.create_random_tensors_hijack on the engine (mdtile_region_noise) == the oracle restatement of upstream
    tilediffusion.py:486-529 (itself pinned bit-exact to upstream in tests/test_oracle_vs_reference.py)."""
    td, U = plugin.tilediffusion, plugin.utils
    torch.manual_seed(5)
    org = torch.randn(2, 4, 40, 56)
    td.Script.create_random_tensors_original_md = staticmethod(lambda *a, **k: org.to(cuda, dtype))
    try:
        settings = {i: U.BBoxSettings(True, fx, fy, fw, fh, "", "", mode, 0.2, seed) for i, (fx, fy, fw, fh, mode, seed) in enumerate(REGION_NOISE)}
        info = {f"Region {i + 1}": {} for i in settings}
        got = td.Script().create_random_tensors_hijack(settings, info, (4, 40, 56), [1, 2])
    finally:
        del td.Script.create_random_tensors_original_md
    assert got.dtype == dtype and got.device.type == "cuda"
    ref = bo.region_noise(org.to(dtype).float(), REGION_NOISE)
    assert torch.equal(got.float().cpu(), ref.to(dtype).float())
    assert [info[f"Region {i + 1}"]["seed"] for i in range(len(REGION_NOISE))] == [r[5] for r in REGION_NOISE]


def test_stream_copy_is_a_copy(plugin, cuda):
    """mdtile_stream_copy (the measurement floor bench.py prints beside the blend kernel): bit-exact copy, ragged tail of the 4 KiB block
    stride included; misaligned / odd-sized requests are refused."""
    E = plugin.engine
    for n in (4, 1024, 1024 * 1024 + 12, 10 * 1024 * 1024 + 4):
        src = torch.randn(n, device=cuda)
        dst = torch.zeros(n + 8, device=cuda)
        E.stream_copy(src, dst[:n])
        assert torch.equal(dst[:n], src) and not dst[n:].any()
    assert E.lib().mdtile_stream_copy(src.data_ptr() + 4, dst.data_ptr(), 64, None) == E.E_ARG        # misaligned source
    assert E.lib().mdtile_stream_copy(src.data_ptr(), dst.data_ptr(), 60, None) == E.E_ARG            # not a multiple of 16 bytes
