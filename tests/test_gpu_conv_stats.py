"""GPU parity of the statistics a conv leaves from its own epilogue (slow mode: the producer of a POOLED GroupNorm's input,
include/mdtile.h "Slow mode (round 5)"): mdtile_conv2d_gn_stats / mdtile_conv2d_rec_stats against
    * the same conv without statistics: y bit-identical;
    * get_var_mean of that y (upstream scripts/tilevae.py:207-215) in fp64 on the CPU, and mdtile_gn_stats (the pass they replace).
Tolerance: a lane adds its <= 16 values in fp32 (~1e-7 relative per lane partial, independent between the lanes), everything above the
lane is fp64: mean within 1e-6 of (|mean| + std), var within 1e-5 relative + 1e-8 of E[x^2] (the E[x^2] - mean^2 form under
cancellation) -- orders under what the slow-mode decode tests allow."""
import pytest
import torch
import torch.nn.functional as F

from hostsim import ldm_decoder as ld
from oracle import gpu_reference as gr

pytestmark = pytest.mark.gpu


def _coef(B, C, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(B, 1, C, generator=g) * 1.5 + 0.25
    s = torch.randn(B, 1, C, generator=g) * 0.5
    return torch.cat([a, s], dim=1).contiguous()


def _var_mean64(y: torch.Tensor, groups: int = 32):
    B, C = y.shape[:2]
    t = y.double().reshape(B, groups, -1)
    return t.var(dim=2, unbiased=False).reshape(-1), t.mean(dim=2).reshape(-1)


def _check(var, mean, y, what):
    v64, m64 = _var_mean64(y.cpu())
    var, mean = var.cpu().double(), mean.cpu().double()
    std = v64.sqrt()
    em = ((mean - m64).abs() / (m64.abs() + std + 1e-30)).max().item()
    ev = ((var - v64).abs() / (1e-5 * v64 + 1e-8 * (v64 + m64 * m64) + 1e-30)).max().item()
    print(f"{what}: mean err {em:.2e} of (|mean| + std), var err {ev:.2f} of its bound (largest mean^2 / var {(m64 * m64 / v64).max().item():.0f})")
    assert em < 1e-6 and ev < 1.0, f"{what}: mean {em} var {ev}"


GN_CASES = [  # B, cin, cout, H, W, residual, output offset (a large mean against the spread: the E[x^2] - mean^2 form under cancellation)
    (1, 128, 128, 8, 32, False, 0.0),        # exactly one block
    (1, 128, 128, 70, 90, True, 0.0),        # ragged rows and columns, 4 couts per group: a quad is a group
    (2, 256, 256, 40, 64, True, 0.0),        # batch, 8 couts per group (both halves of a wave), two cout blocks
    (1, 256, 128, 33, 100, False, 30.0),     # mean >> std
    (1, 512, 512, 24, 40, True, 0.0),        # 16 couts per group, four cout blocks
    (1, 128, 128, 300, 290, False, 0.0),     # > 64 x 256 blocks: several strides of the combine kernel
]


@pytest.mark.parametrize("B,cin,cout,H,W,res,offset", GN_CASES)
def test_handover_conv_leaves_the_statistics_of_its_output(plugin, cuda, B, cin, cout, H, W, res, offset):
    E = plugin.engine
    torch.manual_seed(cin + cout + H)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.bias += offset
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    assert pc.leaves_stats(32)
    x = (torch.randn(B, cin, H, W) * 1.5 + 0.2).to(cuda)
    coef = _coef(B, cin, 5).to(cuda)
    r = torch.randn(B, cout, H, W).to(cuda) if res else None
    y0 = pc(x, residual=r, pre_gn=coef)
    y1, (var, mean) = pc.call_stats(x, coef, residual=r, groups=32)
    assert torch.equal(y0, y1), "the statistics kernel must not change y"
    _check(var, mean, y1, f"hand-over {cin}->{cout} {H}x{W}")
    v2, m2 = E.gn_stats(y1, 32)
    _check(v2, m2, y1, "  the statistics pass on the same y")


REC_CASES = [  # B, cin, cout, H, W (output), upsample, residual, family (0 = the launcher's choice: few item rounds -> two-blocks family + statistics pass inside the call)
    (1, 128, 128, 16, 32, False, False, 1),      # exactly one item
    (1, 512, 512, 70, 90, False, True, 1),       # ragged rows and columns, 16 couts per group, four cout blocks
    (2, 256, 256, 40, 64, False, True, 1),       # batch
    (1, 512, 256, 50, 100, False, False, 1),     # 8 couts per group
    (1, 256, 256, 64, 96, True, False, 1),       # sub-pixel upsample kernel: two pixels per lane, row parities
    (1, 512, 512, 74, 100, True, False, 1),      # upsample, ragged input tiles (37 x 50)
    (2, 128, 128, 36, 80, True, False, 1),       # upsample + batch, 4 couts per group
    (1, 512, 512, 70, 90, False, True, 0),       # the launcher's choice on a small launch
    (1, 256, 256, 600, 560, False, False, 0),    # ... and on one that fills the chip several times (epilogue statistics)
    (1, 256, 256, 640, 512, True, False, 0),
]


@pytest.mark.parametrize("B,cin,cout,H,W,up,res,family", REC_CASES)
def test_record_conv_leaves_the_statistics_of_its_output(plugin, cuda, B, cin, cout, H, W, up, res, family):
    E = plugin.engine
    torch.manual_seed(cin + cout + H + int(up))
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    assert pc.leaves_stats(32, upsample2x=up, rec=True)
    hin, win = (H // 2, W // 2) if up else (H, W)
    x = (torch.randn(B, cin, hin, win) * 1.5 + 0.2).to(cuda)
    xrec = E.rec_from_f32(x, _coef(B, cin, 5).to(cuda))
    r = torch.randn(B, cout, H, W).to(cuda) if res else None
    fam = E.CONV_REC_ONE_BLOCK if family == 1 else 0
    y0, _ = pc.call_rec(xrec, residual=r, upsample2x=up, want_f32=True, want_rec=False, family=fam)
    y1, (var, mean) = pc.call_rec_stats(xrec, residual=r, upsample2x=up, family=fam, groups=32)
    assert torch.equal(y0, y1), "the statistics kernel must not change y"
    _check(var, mean, y1, f"record {'upconv' if up else 'conv'} {cin}->{cout} {H}x{W} family {family}")


def test_narrow_convs_say_no(plugin, cuda):
    E = plugin.engine
    conv = torch.nn.Conv2d(64, 64, 3, padding=1)
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    assert not pc.leaves_stats(32)                      # 64-cout blocks: no statistics kernel (2 couts per group do not fill a quad)
    conv = torch.nn.Conv2d(128, 128, 3, padding=1)
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    assert pc.leaves_stats(32) and not pc.leaves_stats(64) and pc.leaves_stats(8)      # 2 couts per group: no; 16 per group: yes


@pytest.mark.parametrize("size,tile", [(96, 64), (150, 64)], ids=["2x2", "3x3"])
def test_slow_mode_decode_with_and_without_epilogue_statistics(plugin, cuda, size, tile):
    """The whole slow-mode decode (scripts/tilevae.py: lockstep loop) with the statistics from the epilogues against the same decode with a
    statistics pass at every norm (MDTILE_SLOW_STATS=0), and both against the oracle on the GPU.  Full-width SD decoder."""
    tv = plugin.tilevae
    torch.manual_seed(size)
    z = torch.randn(1, 4, size, size)
    dec = ld.make_decoder(11).to(cuda)
    dec.original_forward = dec.forward
    ref = gr.tiled_forward_gpu(dec, z, tile, False).cpu()
    hook = tv.VAEHook(dec, tile, is_decoder=True, fast_decoder=False, fast_encoder=False, color_fix=False)
    outs = {}
    old = tv.SLOW_STATS
    try:
        for on in (True, False):
            tv.SLOW_STATS = on
            outs[on] = hook(z.to(cuda)).cpu()
    finally:
        tv.SLOW_STATS = old
    scale = ref.abs().max().item()
    e_on, e_off = (outs[True] - ref).abs().max().item() / scale, (outs[False] - ref).abs().max().item() / scale
    d = (outs[True] - outs[False]).abs().max().item() / scale
    print(f"slow mode {size}/{tile}: vs oracle {e_on:.2e} (epilogue statistics) / {e_off:.2e} (statistics pass); between them {d:.2e}")
    assert e_on < 2e-4 and e_off < 2e-4 and d < 2e-5
