"""GPU parity of the record-image conv path (csrc/vae_conv_rec.hip): fp32 -> record image (+ fused fixed-statistics GroupNorm +
SiLU), the record 3x3 conv and the record sub-pixel upsample conv (fp32 and record outputs, residual, ragged tiles, batch),
and the whole fast-mode tiled decode with the record hand-over against the fp32 hand-over and the oracle.
Tolerances: split-bf16 operands carry 16 significand bits per factor -> <= 1e-4 of the output range per conv (fp32 accumulate);
a record image reproduces its fp32 source to 2^-16 relative per element."""
import pytest
import torch
import torch.nn.functional as F

from hostsim import ldm_decoder as ld
from oracle import vae_oracle as vo

pytestmark = pytest.mark.gpu


def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _coef(B, C, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(B, 1, C, generator=g) * 1.5 + 0.25
    s = torch.randn(B, 1, C, generator=g) * 0.5
    return torch.cat([a, s], dim=1).contiguous()        # [B, 2, C] = (a, s)


def _act(x, coef):
    B, C = x.shape[:2]
    return F.silu(x * coef[:, 0].view(B, C, 1, 1) + coef[:, 1].view(B, C, 1, 1))


@pytest.mark.parametrize("B,C,H,W", [(1, 32, 5, 7), (2, 128, 17, 45), (1, 512, 24, 40), (1, 64, 70, 300)])
def test_record_image_roundtrip(plugin, cuda, B, C, H, W):
    E = plugin.engine
    torch.manual_seed(C + H)
    x = torch.randn(B, C, H, W) * 2.0 + 0.3
    back = E.rec_from_f32(x.to(cuda)).to_f32().cpu()
    assert (back - x).abs().max().item() <= 2.0 ** -15 * x.abs().max().item()
    assert ((back - x).abs() <= 2.0 ** -16 * x.abs() + 1e-30).all()      # hi + lo: 16 significand bits per element
    coef = _coef(B, C, 3)
    ref = _act(x, coef)
    got = E.rec_from_f32(x.to(cuda), coef.to(cuda)).to_f32().cpu()
    assert _rel(got, ref) < 2e-5


REC_CASES = [  # B, cin, cout, H, W (output), upsample, residual
    (1, 128, 128, 16, 32, False, False),     # exactly one block
    (1, 128, 128, 17, 45, False, True),      # ragged rows and columns
    (2, 256, 128, 40, 36, False, True),      # batch 2, 16 K-steps
    (1, 512, 512, 24, 40, False, True),      # SD mid-block width, 4 cout blocks
    (1, 96, 128, 33, 70, False, False),      # 6 K-steps, three pixel-tile columns
    (1, 256, 256, 50, 100, False, True),     # many pixel tiles x 2 cout blocks (block -> XCD mapping)
    (1, 128, 128, 32, 48, True, False),      # sub-pixel upsample: NK = 8 (8 % 3 = 2: skipped K-steps in the last trip)
    (2, 256, 128, 40, 36, True, True),       # upsample + residual + batch (NK = 16)
    (1, 512, 512, 74, 100, True, False),     # upsample, ragged input tiles (37 x 50), 4 cout blocks (NK = 32)
    (1, 96, 128, 18, 66, True, True),        # upsample, NK = 6 (whole trips only)
]


@pytest.mark.parametrize("B,cin,cout,H,W,up,res", REC_CASES)
def test_conv_rec_vs_torch(plugin, cuda, B, cin, cout, H, W, up, res):
    E = plugin.engine
    torch.manual_seed(cin * 7 + cout + H)
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1)
    hin, win = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(B, cin, hin, win)
    in_coef = None if up else _coef(B, cin, 5)          # upsample.conv reads the raw activation, conv1 / conv2 a normalised one
    out_coef = _coef(B, cout, 9)
    with torch.no_grad():
        xin = x if in_coef is None else _act(x, in_coef)
        ref = conv(F.interpolate(xin, scale_factor=2.0, mode="nearest") if up else xin)
        r = torch.randn_like(ref) if res else None
        if res:
            ref = ref + r
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    assert pc.takes_rec(up)
    xrec = E.rec_from_f32(x.to(cuda), None if in_coef is None else in_coef.to(cuda))
    rr = None if r is None else r.to(cuda)
    # fp32 + activated record output in one launch
    y, yrec = pc.call_rec(xrec, residual=rr, upsample2x=up, want_f32=True, want_rec=True, rec_coef=out_coef.to(cuda))
    err = _rel(y.cpu(), ref)
    assert err < 5e-5, f"record conv fp32 output: rel err {err}"
    got = yrec.to_f32().cpu()
    want = _act(ref, out_coef)
    assert _rel(got, want) < 2e-4, f"record conv activated record output: rel err {_rel(got, want)}"
    # the zero border is part of the image: a second conv reading it must see 'same' zero padding
    conv2 = torch.nn.Conv2d(cout, 128, 3, 1, 1)
    pc2 = E.PackedConv(conv2.weight.detach().to(cuda), conv2.bias.detach().to(cuda))
    y2, _ = pc2.call_rec(yrec, want_f32=True)
    with torch.no_grad():
        ref2 = conv2(want)
    assert _rel(y2.cpu(), ref2) < 2e-4, f"chained record conv: rel err {_rel(y2.cpu(), ref2)}"
    # raw record output only (what conv2 hands to upsample.conv), no fp32 copy
    y3, yrec3 = pc.call_rec(xrec, residual=rr, upsample2x=up, want_f32=False, want_rec=True)
    assert y3 is None
    assert _rel(yrec3.to_f32().cpu(), ref) < 5e-5


@pytest.mark.parametrize("cin,cout,H,W,B", [(128, 3, 40, 70, 1), (128, 8, 17, 33, 2), (256, 16, 24, 40, 1)])
def test_conv_rec_narrow_output(plugin, cuda, cin, cout, H, W, B):
    """conv_out (cout 3 decoder / 8 encoder) on the record path: norm_out + SiLU arrive fused in the record image, one 32-cout tile per
    block, fp32 output for the real couts only."""
    E = plugin.engine
    torch.manual_seed(cin + cout)
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1)
    x = torch.randn(B, cin, H, W)
    coef = _coef(B, cin, 4)
    with torch.no_grad():
        ref = conv(_act(x, coef))
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    assert pc.takes_rec()
    y, yr = pc.call_rec(E.rec_from_f32(x.to(cuda), coef.to(cuda)), want_f32=True)
    assert yr is None and y.shape == ref.shape
    assert _rel(y.cpu(), ref) < 5e-5


def test_conv_rec_matches_fp32_handover_kernel(plugin, cuda):
    """Same arithmetic contract as the fused-GroupNorm split-bf16 kernel (vae_conv_bf16x3.hip): the two families agree to fp32
    round-off of exp / rcp, far below the split's own 2^-16."""
    E = plugin.engine
    torch.manual_seed(4)
    B, cin, cout, H, W = 1, 256, 256, 37, 61
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1)
    x = (torch.randn(B, cin, H, W) * 1.3).to(cuda)
    coef = _coef(B, cin, 2).to(cuda)
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    a = pc(x, pre_gn=coef)
    b, _ = pc.call_rec(E.rec_from_f32(x, coef), want_f32=True)
    assert _rel(b.cpu(), a.cpu()) < 2e-5


@pytest.mark.parametrize("fast", [True])
def test_tiled_decode_record_path_vs_fp32_handover_and_oracle(plugin, cuda, fast):
    dec_cpu = ld.make_decoder(3)
    torch.manual_seed(5)
    z = torch.randn(1, 4, 34, 42)
    ref = vo.tiled_forward(dec_cpu, z, 12, fast)
    dec = ld.make_decoder(3).to(cuda)
    dec.original_forward = dec.forward
    outs = {}
    old = plugin.tilevae.REC_PATH
    try:
        for rec in (True, False):
            plugin.tilevae.REC_PATH = rec
            hook = plugin.tilevae.VAEHook(dec, 12, is_decoder=True, fast_decoder=fast, fast_encoder=False, color_fix=False)
            outs[rec] = hook(z.to(cuda)).cpu()
    finally:
        plugin.tilevae.REC_PATH = old
    assert _rel(outs[True], ref) < 2e-4
    assert _rel(outs[False], ref) < 2e-4
    assert _rel(outs[True], outs[False]) < 5e-5


def test_tiled_encode_record_path(plugin, cuda):
    """Encoder direction through the same executor: stride-2 Downsample convs read fp32, everything else records."""
    enc_cpu = ld.make_encoder(2)
    torch.manual_seed(6)
    x = torch.randn(1, 3, 168, 136)
    ref = vo.tiled_forward(enc_cpu, x, 64, True, is_decoder=False, color_fix=False)
    enc = ld.make_encoder(2).to(cuda)
    enc.original_forward = enc.forward
    hook = plugin.tilevae.VAEHook(enc, 64, is_decoder=False, fast_decoder=False, fast_encoder=True, color_fix=False)
    out = hook(x.to(cuda)).cpu()
    assert _rel(out, ref) < 2e-4


# ---- live-window narrowing (mdtile_upconv2d_rec_window; scripts/tilevae.py: live_windows) ---------------------------------------------
WINDOW_CASES = [  # B, cin, cout, Hin, Win, (y0, x0, h, w)
    (1, 128, 128, 40, 70, (4, 4, 30, 60)),        # interior window: all four edges read the image's own neighbours
    (1, 128, 128, 40, 70, (0, 0, 33, 41)),        # anchored top-left: two edges are the image's zero border
    (2, 256, 128, 37, 50, (5, 9, 32, 41)),        # ends on the bottom / right edge of the image, batch 2, ragged input tiles
    (1, 512, 512, 24, 40, (3, 6, 9, 31)),         # 4 cout blocks, window smaller than one block row
    (1, 96, 128, 19, 66, (1, 1, 17, 64)),         # NK = 6
    (1, 256, 256, 64, 64, (0, 0, 64, 64)),        # the whole image as a window == the plain call
]


@pytest.mark.parametrize("B,cin,cout,Hin,Win,win", WINDOW_CASES)
def test_upconv_window_equals_the_same_pixels_of_the_whole_image_call(plugin, cuda, B, cin, cout, Hin, Win, win):
    """Bit for bit: the arithmetic of an output pixel does not depend on where its block sits."""
    E = plugin.engine
    torch.manual_seed(cin + Hin)
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1)
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    x = torch.randn(B, cin, Hin, Win).to(cuda)
    coef = _coef(B, cout, 5).to(cuda)
    xr = E.rec_from_f32(x)
    y_full, r_full = pc.call_rec(xr, upsample2x=True, want_f32=True, want_rec=True, rec_coef=coef)
    y0, x0, h, w = win
    y_win, r_win = pc.call_rec(xr, upsample2x=True, want_f32=True, want_rec=True, rec_coef=coef, window=win)
    assert y_win.shape == (B, cout, 2 * h, 2 * w) and r_win.shape == (B, cout, 2 * h, 2 * w)
    assert torch.equal(y_win, y_full[:, :, 2 * y0:2 * (y0 + h), 2 * x0:2 * (x0 + w)])
    assert torch.equal(r_win.to_f32(), r_full.to_f32()[:, :, 2 * y0:2 * (y0 + h), 2 * x0:2 * (x0 + w)])
    # the record output is a well-formed image of its own: the same records as the whole-image call inside, a ZERO border around
    # (the next conv's padding -- not the neighbours the whole image has there)
    dw = r_win.records()          # [B, 2, cout / 8, 2 h + 2, 2 w + 2, 4]: the logical image incl. its border records
    df = r_full.records()
    assert torch.equal(dw[:, :, :, 1:-1, 1:-1], df[:, :, :, 1 + 2 * y0:1 + 2 * (y0 + h), 1 + 2 * x0:1 + 2 * (x0 + w)])
    assert not dw[:, :, :, 0].any() and not dw[:, :, :, -1].any() and not dw[:, :, :, :, 0].any() and not dw[:, :, :, :, -1].any()
    # and against torch fp32
    with torch.no_grad():
        ref = conv.to(cuda)(F.interpolate(x, scale_factor=2.0, mode="nearest"))[:, :, 2 * y0:2 * (y0 + h), 2 * x0:2 * (x0 + w)]
    assert _rel(y_win, ref) < 5e-5


def test_upconv_window_origin_per_image(plugin, cuda):
    """Stacked tiles of one shape keep their own window origin (left- and right-edge tiles of an image): image b of the launch reads
    the window at (y0[b], x0[b]); 12 images with origins that repeat every 8 are taken too, other patterns beyond 8 are refused."""
    E = plugin.engine
    torch.manual_seed(4)
    conv = torch.nn.Conv2d(128, 128, 3, 1, 1)
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    B, Hin, Win, h, w = 5, 30, 44, 21, 35
    x = torch.randn(B, 128, Hin, Win).to(cuda)
    xr = E.rec_from_f32(x)
    y_full, _ = pc.call_rec(xr, upsample2x=True)
    y0s, x0s = [0, 9, 4, 0, 7], [9, 0, 3, 0, 9]
    y_win, r_win = pc.call_rec(xr, upsample2x=True, want_rec=True, window=(y0s, x0s, h, w))
    for b in range(B):
        assert torch.equal(y_win[b], y_full[b, :, 2 * y0s[b]:2 * (y0s[b] + h), 2 * x0s[b]:2 * (x0s[b] + w)]), f"image {b}"
    assert torch.equal(r_win.to_f32(), y_win) or _rel(r_win.to_f32(), y_win) < 2.0 ** -15
    x12 = torch.randn(12, 128, 12, 40).to(cuda)
    xr12 = E.rec_from_f32(x12)
    f12, _ = pc.call_rec(xr12, upsample2x=True)
    o = [0, 1, 2, 3, 4, 3, 2, 1]
    w12, _ = pc.call_rec(xr12, upsample2x=True, window=([o[b & 7] for b in range(12)], [2 * o[b & 7] for b in range(12)], 8, 32))
    for b in range(12):
        assert torch.equal(w12[b], f12[b, :, 2 * o[b & 7]:2 * o[b & 7] + 16, 4 * o[b & 7]:4 * o[b & 7] + 64])
    with pytest.raises(E.MdtileError):
        pc.call_rec(xr12, upsample2x=True, window=([b % 3 for b in range(12)], [0] * 12, 8, 32))


def test_upconv_window_rejects_a_window_outside_the_image(plugin, cuda):
    E = plugin.engine
    conv = torch.nn.Conv2d(128, 128, 3, 1, 1)
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    xr = E.rec_from_f32(torch.randn(1, 128, 16, 16, device=cuda))
    for bad in ((0, 0, 17, 16), (-1, 0, 8, 8), (4, 10, 8, 8), (0, 0, 0, 4)):
        with pytest.raises(E.MdtileError):
            pc.call_rec(xr, upsample2x=True, window=bad)


@pytest.mark.parametrize("hw,ts,N", [((64, 88), 32, 1), ((40, 40), 64, 1), ((70, 40), 16, 2), ((70, 40), 16, 3)])
def test_fast_decode_with_live_windows_equals_the_whole_tile_sweep(plugin, cuda, hw, ts, N):
    """Full-width SD decoder, fast mode: tiles narrowed where the resolution doubles (default) == whole padded tiles, BIT FOR BIT on the
    assembled image (tiles of 54^2 / 54 x 50 latent px incl. image-edge tiles; one tile = no padding = nothing to shed); the narrowed
    sweep issues smaller launches.  N = 2 / 3 latents per call: a stacked sweep holds T x N images, each with the window origin of its tile
    (at most 8 origins per launch: 3 tiles x 2, 2 tiles x 3)."""
    tv = plugin.tilevae
    dec = ld.make_decoder(5).to(cuda)
    dec.original_forward = dec.forward
    torch.manual_seed(hw[0])
    z = torch.randn(N, 4, *hw).to(cuda)
    outs, px = {}, {}
    orig = plugin.engine.PackedConv.call_rec
    old = tv.LIVE_WINDOW
    try:
        for live in (True, False):
            tv.LIVE_WINDOW = live
            count = [0, 0]

            def counted(self, x, *a, **kw):
                out = orig(self, x, *a, **kw)          # always (y, yr) (the statistics-leaving calls are call_rec_stats / call_stats)
                y, yr = out
                o = y if y is not None else yr
                count[0] += o.shape[0] * o.shape[1] * o.shape[2] * o.shape[3]
                count[1] += 1 if kw.get("window") else 0
                return out

            plugin.engine.PackedConv.call_rec = counted
            hook = tv.VAEHook(dec, ts, is_decoder=True, fast_decoder=True, fast_encoder=False, color_fix=False)
            outs[live] = hook(z).cpu()
            px[live] = tuple(count)
    finally:
        tv.LIVE_WINDOW = old
        plugin.engine.PackedConv.call_rec = orig
    assert torch.equal(outs[True], outs[False])
    if ts < min(hw):
        assert px[True][1] > 0 and px[False][1] == 0 and px[True][0] < 0.9 * px[False][0], px
    else:
        assert px[True] == px[False]


@pytest.mark.parametrize("mode", ["f32_engine", "narrow_decoder"])
def test_live_windows_on_the_fp32_handover_kernels(plugin, cuda, mode):
    """Upsample convs that the record kernels do not take -- the exact-fp32 engine (mdtile_set_precision; the bench's `value_f32`) and a decoder
    whose widths are not multiples of 128 -- get their window from the fp32 hand-over kernel over window + halo (VAEHook._upconv_window_f32):
    the assembled image is again bit-identical to the whole-tile sweep, and the windows are taken (smaller conv outputs)."""
    E, tv = plugin.engine, plugin.tilevae
    dec = (ld.make_decoder(6) if mode == "f32_engine" else ld.make_decoder(6, small=True)).to(cuda)
    dec.original_forward = dec.forward
    torch.manual_seed(9)
    z = torch.randn(1, 4, 56, 70).to(cuda)
    outs, px = {}, {}
    orig = E.PackedConv.__call__
    old = tv.LIVE_WINDOW
    try:
        if mode == "f32_engine":
            E.set_precision(E.PRECISION_F32)
        for live in (True, False):
            tv.LIVE_WINDOW = live
            count = [0]

            def counted(self, x, *a, **kw):
                y = orig(self, x, *a, **kw)            # always the fp32 tensor
                count[0] += y.numel() if self.ksize == 3 else 0
                return y

            E.PackedConv.__call__ = counted
            hook = tv.VAEHook(dec, 24, is_decoder=True, fast_decoder=True, fast_encoder=False, color_fix=False)
            outs[live] = hook(z).cpu()
            px[live] = count[0]
    finally:
        tv.LIVE_WINDOW = old
        E.PackedConv.__call__ = orig
        E.set_precision(E.PRECISION_BF16X3)
    assert torch.equal(outs[True], outs[False])
    assert 0 < px[True] < 0.92 * px[False], px


@pytest.mark.parametrize("B,cin,cout,H,W,up,res", REC_CASES + [(3, 128, 256, 45, 77, False, True), (2, 512, 128, 30, 200, True, False), (1, 64, 128, 90, 130, False, True)])
def test_two_blocks_per_cu_kernels_are_bit_identical_to_the_one_block_kernels(plugin, cuda, B, cin, cout, H, W, up, res):
    """csrc/vae_conv_rec2.hip (two independent 4-wave blocks per CU) issues every accumulator's MFMAs in the order of
    csrc/vae_conv_rec.hip (one 8-wave block per CU): same fp32 output, same record image, bit for bit -- also with many items per
    block (the ring of step chunks and the input stages run on across item boundaries).  The family is named by the call's flags
    (MDTILE_CONV_REC_ONE_BLOCK / _TWO_BLOCKS); the start-up skew variants are probe switches of the PROBES build only."""
    E = plugin.engine
    torch.manual_seed(cin + 3 * cout + H)
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1)
    hin, win = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(B, cin, hin, win)
    out_coef = _coef(B, cout, 11).to(cuda)
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    xrec = E.rec_from_f32(x.to(cuda), None if up else _coef(B, cin, 5).to(cuda))
    rr = torch.randn(B, cout, H, W).to(cuda) if res else None
    y1, r1 = pc.call_rec(xrec, residual=rr, upsample2x=up, want_f32=True, want_rec=True, rec_coef=out_coef, family=E.CONV_REC_ONE_BLOCK)
    for _ in range(2):      # twice: the per-CU arrival counters of the start-up skew carry over from launch to launch
        y2, r2 = pc.call_rec(xrec, residual=rr, upsample2x=up, want_f32=True, want_rec=True, rec_coef=out_coef, family=E.CONV_REC_TWO_BLOCKS)
        assert torch.equal(y1, y2), f"fp32 output differs: {_rel(y2, y1)}"
        assert torch.equal(r1.records(), r2.records()), "record output differs"


@pytest.mark.parametrize("cin,cout,H,W,up,res", [(128, 128, 1200, 1056, False, True), (128, 128, 1200, 1056, False, False), (64, 128, 1088, 1056, True, False)])
def test_many_items_per_block(plugin, cuda, cin, cout, H, W, up, res):
    """Launches of more than six item rounds per CU (the persistent loop runs long: ring slots, input stages and -- for a conv2 -- the
    residual rows that the epilogue of item i loads into the accumulators of item i + 1 all run on across item boundaries), against torch
    fp32, one-block family named explicitly."""
    E = plugin.engine
    torch.manual_seed(cin + cout + H)
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1)
    hin, win = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(1, cin, hin, win)
    out_coef = _coef(1, cout, 11).to(cuda)
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    xrec = E.rec_from_f32(x.to(cuda), None)
    rr = torch.randn(1, cout, H, W).to(cuda) if res else None
    y0, r0 = pc.call_rec(xrec, residual=rr, upsample2x=up, want_f32=True, want_rec=True, rec_coef=out_coef, family=E.CONV_REC_ONE_BLOCK)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xin.to(cuda), conv.weight.detach().to(cuda), conv.bias.detach().to(cuda), padding=1)
    if res:
        ref = ref + rr
    assert _rel(y0, ref) <= 5e-5
    assert _rel(r0.to_f32(), _act(ref, out_coef)) <= 5e-5


@pytest.mark.parametrize("blocks", [4, 8], ids=["one_block_per_cu", "two_blocks_per_cu"])
def test_upconv_windows_under_both_kernel_families(plugin, cuda, blocks):
    """mdtile_upconv2d_rec_window (per-image window origins inside a larger input image, live-window narrowing of the decoder tiles) through
    the one-block kernel AND the two-blocks-per-CU kernel (csrc/vae_conv_rec2.hip: k_upconv_rec2 tiles the window in 4-row items and reads
    the whole image's pitch): every image's window equals the same pixels of the whole-image call, fp32 and records, bit for bit."""
    E = plugin.engine
    assert (E.CONV_REC_ONE_BLOCK, E.CONV_REC_TWO_BLOCKS) == (4, 8)
    torch.manual_seed(8)
    conv = torch.nn.Conv2d(256, 128, 3, 1, 1)
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    B, Hin, Win, h, w = 4, 37, 52, 22, 33
    x = torch.randn(B, 256, Hin, Win).to(cuda)
    xr = E.rec_from_f32(x)
    coef = _coef(B, 128, 3).to(cuda)
    y_full, r_full = pc.call_rec(xr, upsample2x=True, want_f32=True, want_rec=True, rec_coef=coef, family=blocks)
    y0s, x0s = [0, 15, 6, 11], [19, 0, 7, 13]
    y_win, r_win = pc.call_rec(xr, upsample2x=True, want_f32=True, want_rec=True, rec_coef=coef, window=(y0s, x0s, h, w), family=blocks)
    rf, rw = r_full.records(), r_win.records()
    for b in range(B):
        ys_, xs_ = slice(2 * y0s[b], 2 * (y0s[b] + h)), slice(2 * x0s[b], 2 * (x0s[b] + w))
        assert torch.equal(y_win[b], y_full[b, :, ys_, xs_]), f"fp32, image {b}"
        assert torch.equal(rw[b, :, :, 1:-1, 1:-1], rf[b, :, :, 1 + ys_.start:1 + ys_.stop, 1 + xs_.start:1 + xs_.stop]), f"records, image {b}"
    assert not rw[:, :, :, 0].any() and not rw[:, :, :, -1].any() and not rw[:, :, :, :, 0].any() and not rw[:, :, :, :, -1].any()
    with torch.no_grad():
        ref = conv.to(cuda)(F.interpolate(x, scale_factor=2.0, mode="nearest"))
    assert _rel(y_full, ref) < 5e-5
