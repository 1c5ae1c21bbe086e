"""Bit-identity check of the dripped-epilogue record conv (probes/csrc/vae_conv_recd.hip) against the shipping one-block kernel.  The
kernel was built, measured and rejected in round 5 (docs/history/r5.md); since round 6 it lives in the PROBES twin of the library only
(python -m mdtile.build --probes), so this check is a probe, not part of tests/:
        python -m pytest probes/check_conv_drip.py -q          (on the GPU box)"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"), os.path.join(ROOT, "probes")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.skipif(not torch.cuda.is_available(), reason="needs the MI355X")


@pytest.fixture(scope="module")
def plugin():
    import mdtile as E
    import _probes_lib
    _probes_lib.use(E)

    class P:
        engine = E
    return P


@pytest.fixture(scope="module")
def cuda():
    return torch.device("cuda:0")


def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)



def _coef(B, C, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(B, 1, C, generator=g) * 1.5 + 0.25
    s = torch.randn(B, 1, C, generator=g) * 0.5
    return torch.cat([a, s], dim=1).contiguous()        # [B, 2, C] = (a, s)



DRIP_CASES = [  # B, cin, cout, H, W, residual, want_f32, coef
    (1, 128, 128, 16, 32, False, True, True),        # one pixel tile: two 64-cout items, one per block -- only the un-dripped final epilogue runs
    (1, 128, 128, 17, 45, True, True, True),         # ragged rows and columns (rows past H: slots that issue no stores -> the uncounted wait)
    (2, 256, 128, 40, 36, True, True, True),         # batch 2, NK = 16 (two trips after the slot trip)
    (1, 512, 512, 24, 40, True, True, True),         # 8 cout blocks of 64, NK = 32
    (1, 128, 128, 1200, 1056, True, True, True),     # ~10 items per block: conv2 (fp32 + records + residual), every slot kind, many swaps
    (1, 128, 128, 1200, 1056, False, False, True),   # conv1: records only (no fp32 stores, zero start values)
    (1, 256, 128, 700, 1000, False, True, False),    # fp32 + raw records (no activation), no residual
    (1, 128, 256, 333, 517, True, True, True),       # odd sizes, 4 cout blocks
    (3, 128, 128, 278, 278, True, False, True),      # three stacked tiles as the 8K decode launches them; records + residual, no fp32
]


@pytest.mark.parametrize("B,cin,cout,H,W,res,f32,act", DRIP_CASES)
def test_dripped_epilogue_kernel_is_bit_identical_to_the_one_block_kernel(plugin, cuda, B, cin, cout, H, W, res, f32, act):
    """probes/csrc/vae_conv_recd.hip (64-cout items, two accumulator sets per wave, the previous item's epilogue issued in slots between the
    K-steps of the running one; MDTILE_CONV_REC_DRIP) against csrc/vae_conv_rec.hip's one-block kernel: every accumulator sees the same
    MFMAs in the same order from the same start value, so fp32 output and record image agree bit for bit -- also across many item
    boundaries (register-set swaps, residual rows loaded into the sealed set, counted vmcnt waits)."""
    E = plugin.engine
    torch.manual_seed(cin + 3 * cout + H)
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1)
    x = torch.randn(B, cin, H, W)
    out_coef = _coef(B, cout, 11).to(cuda) if act else None
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    xrec = E.rec_from_f32(x.to(cuda), _coef(B, cin, 5).to(cuda))
    rr = torch.randn(B, cout, H, W).to(cuda) if res else None
    y1, r1 = pc.call_rec(xrec, residual=rr, want_f32=f32, want_rec=True, rec_coef=out_coef, family=E.CONV_REC_ONE_BLOCK)
    for _ in range(2):
        y2, r2 = pc.call_rec(xrec, residual=rr, want_f32=f32, want_rec=True, rec_coef=out_coef, family=E.CONV_REC_DRIP)
        if f32:
            assert torch.equal(y1, y2), f"fp32 output differs: {_rel(y2, y1)}"
        assert torch.equal(r1.records(), r2.records()), "record output differs"
    if f32 and H * W < 200000:
        ref = F.conv2d(x.to(cuda), conv.weight.detach().to(cuda), conv.bias.detach().to(cuda), padding=1) if not act else None
        if ref is not None:
            assert _rel(y2, ref + rr if res else ref) <= 5e-5


def test_dripped_epilogue_fp32_only_output(plugin, cuda):
    """fp32 output alone (no record image): only the A half of the slots issues stores."""
    E = plugin.engine
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(128, 128, 3, 1, 1)
    x = torch.randn(1, 128, 600, 800)
    pc = E.PackedConv(conv.weight.detach().to(cuda), conv.bias.detach().to(cuda))
    xrec = E.rec_from_f32(x.to(cuda))
    rr = torch.randn(1, 128, 600, 800).to(cuda)
    y1, _ = pc.call_rec(xrec, residual=rr, want_f32=True, want_rec=False, family=E.CONV_REC_ONE_BLOCK)
    y2, _ = pc.call_rec(xrec, residual=rr, want_f32=True, want_rec=False, family=E.CONV_REC_DRIP)
    assert torch.equal(y1, y2)
