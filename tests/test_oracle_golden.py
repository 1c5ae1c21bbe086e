"""Oracle vs the committed golden vectors (tests/golden/*.npz, produced by the upstream code -- make_golden.py).
Runs everywhere, no GPU and no /root/reference needed."""
import numpy as np
import pytest
import torch

from oracle import blend_oracle as bo
from hostsim import ldm_decoder as ld
from oracle import vae_oracle as vo


def test_maps(golden_maps):
    for key in golden_maps.files:
        kind, dims = key.split("_", 1)
        if kind == "gauss":
            tw, th = map(int, dims.split("x"))
            got = bo.gaussian_weights(tw, th)
        else:
            wh, r = dims.rsplit("_", 1)
            w, h = map(int, wh.split("x"))
            got = bo.feather_mask(w, h, float(r))
        assert np.array_equal(got.numpy(), golden_maps[key]), key


def test_grids(cases):
    for g in cases["grid"]:
        w, h, tw, th, ov, bs = g["args"]
        boxes, batches, tw2, th2, ov2 = bo.init_grid(w, h, tw, th, ov, bs)
        assert [list(b) for b in boxes] == g["boxes"]
        assert len(batches) == g["num_batches"] and len(batches[0]) == g["tile_bs"]
        wm = bo.grid_weight_map(w, h, boxes, 1.0)
        assert float(wm.min()) == g["wmin"] and float(wm.max()) == g["wmax"] and float(wm.double().sum()) == g["wsum"]


def test_tiles(cases):
    for t in cases["tiles"]:
        h, w, ts, is_dec = t["args"]
        ins, outs = vo.split_tiles(h, w, ts, is_dec)
        assert ins == t["ins"] and outs == t["outs"]


def _regions(c):
    return [bo.Region(*bo.region_rect(c["W"], c["H"], fx, fy, fw, fh), mode, fr) for (fx, fy, fw, fh, mode, fr) in (c["regions"] or [])]


def test_blend(cases, golden_blend):
    for c in cases["blend"]:
        o = bo.BlendOracle(c["method"], c["W"], c["H"], c["tw"], c["th"], c["ov"], c["bs"], _regions(c), c["bg"])
        torch.manual_seed(c["seed"])
        x = torch.randn(c["N"], 4, c["H"], c["W"])
        out = o.evaluate(x, bo.synthetic_denoiser, bo.synthetic_region_denoiser)
        assert np.array_equal(o.weights.numpy(), golden_blend[c["name"] + "/weights"]), c["name"]
        assert np.array_equal(out.numpy(), golden_blend[c["name"] + "/out"]), c["name"]


def test_vae(cases, golden_vae):
    s = cases["vae_stride"]
    for c in cases["vae"]:
        dec = ld.make_decoder(c["dec_seed"], small=True)
        torch.manual_seed(c["seed"])
        z = torch.randn(1, 4, c["H"], c["W"])
        out = vo.tiled_forward(dec, z, c["ts"], c["fast"])
        assert np.array_equal(out[:, :, ::s, ::s].numpy(), golden_vae[c["name"] + "/sub"]), c["name"]
        mom = np.array([out.double().sum().item(), (out.double() ** 2).sum().item(), out.abs().max().item()])
        assert np.array_equal(mom, golden_vae[c["name"] + "/moments"]), c["name"]


def test_vae_encoder_goldens():
    """Encoder direction (pad 32, Downsample, color_fix): the oracle reproduces the upstream-generated vectors bit for bit."""
    import json, os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(here, "cases_enc.json")) as f:
        enc_cases = json.load(f)["enc"]
    gold = np.load(os.path.join(here, "vae_enc.npz"))
    for c in enc_cases:
        enc = ld.make_encoder(c["enc_seed"], small=True)
        torch.manual_seed(c["seed"])
        x = torch.randn(1, 3, c["H"], c["W"])
        out = vo.tiled_forward(enc, x, c["ts"], c["fast"], is_decoder=False, color_fix=c["color_fix"])
        assert np.array_equal(out.numpy(), gold[c["name"] + "/out"]), c["name"]


def test_gn_attn(golden_vae):
    torch.manual_seed(11)
    t = torch.randn(2, 64, 9, 13) * 3 + 0.5
    var, mean = vo.get_var_mean(t, 32)
    g, b = torch.randn(64), torch.randn(64)
    assert np.array_equal(var.numpy(), golden_vae["gn/var"]) and np.array_equal(mean.numpy(), golden_vae["gn/mean"])
    assert np.array_equal(vo.custom_group_norm(t, 32, mean, var, g, b).numpy(), golden_vae["gn/out"])
    torch.manual_seed(12)
    ab = ld.AttnBlock(64).eval()
    hx = torch.randn(1, 64, 7, 9)
    with torch.no_grad():
        assert np.array_equal(vo.attn_body(ab, hx).numpy(), golden_vae["attn/out"])


def test_only_tiles_is_the_full_sweep_restricted_to_those_tiles():
    """oracle/vae_oracle.py: tiled_forward(only_tiles=...) -- fast mode, every GroupNorm frozen -- returns exactly the rectangles the full
    sweep (pinned to upstream above and in tests/test_oracle_vs_reference.py) writes for those tiles: bench.py uses it to check single
    tiles of the 8K image, which the checker cannot decode whole."""
    from oracle import vae_oracle as vo
    dec = ld.make_decoder(3, small=True)
    torch.manual_seed(1)
    z = torch.randn(1, 4, 70, 58)
    full = vo.tiled_forward(dec, z, 16, True)
    ins, outs = vo.split_tiles(70, 58, 16)
    sel = [0, 4, len(ins) // 2, len(ins) - 1]
    got = vo.tiled_forward(dec, z, 16, True, only_tiles=sel)
    assert [ob for ob, _ in got] == [outs[t] for t in sel]
    for ob, crop in got:
        assert torch.equal(full[:, :, ob[2]:ob[3], ob[0]:ob[1]], crop)
    with pytest.raises(AssertionError):
        vo.tiled_forward(dec, z, 16, False, only_tiles=sel)      # slow mode pools statistics over ALL tiles
