"""CPU model of the engine's split-bf16 ("bf16x3") arithmetic on a WHOLE decoder (tool, no GPU): every conv and both attention
contractions are evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  (optionally + a_lo*b_lo) with fp32 accumulation and compared
with the same network in fp64.  Used to size the error of the default precision on trained-like ("stress") statistics before
spending GPU minutes; tests/test_split_bf16_model.py pins the small cases.

    python tools/bf16x3_model.py [--latent 32] [--logit-std 8] [--terms 3|4] [--default-init]
"""
from __future__ import annotations

import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hostsim import ldm_decoder as ld  # noqa: E402


def split(x: torch.Tensor):
    hi = x.to(torch.bfloat16).to(torch.float32)
    lo = (x - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


class SplitArithmetic:
    """Context manager: F.conv2d / torch.bmm of fp32 tensors become their split-bf16 forms."""

    def __init__(self, terms: int = 3):
        self.terms = terms

    def conv2d(self, x, w, b=None, *a, **k):
        if x.dtype != torch.float32:
            return self._conv(x, w, b, *a, **k)
        xh, xl = split(x)
        wh, wl = split(w)
        y = self._conv(xl, wh, None, *a, **k) + self._conv(xh, wl, None, *a, **k)
        if self.terms == 4:
            y = y + self._conv(xl, wl, None, *a, **k)
        y = y + self._conv(xh, wh, None, *a, **k)
        return y if b is None else y + b.view(1, -1, 1, 1)

    def bmm(self, p, q):
        if p.dtype != torch.float32:
            return self._bmm(p, q)
        ph, pl = split(p)
        qh, ql = split(q)
        y = self._bmm(pl, qh) + self._bmm(ph, ql)
        if self.terms == 4:
            y = y + self._bmm(pl, ql)
        return y + self._bmm(ph, qh)

    def __enter__(self):
        self._conv, self._bmm = F.conv2d, torch.bmm
        F.conv2d, torch.bmm = self.conv2d, self.bmm
        torch.nn.functional.conv2d = self.conv2d
        return self

    def __exit__(self, *exc):
        F.conv2d, torch.bmm = self._conv, self._bmm


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@torch.no_grad()
def run(latent: int, stress, terms: int, seed: int = 3, per_layer: bool = False):
    dec = ld.make_decoder(seed, stress=stress)
    dec64 = ld.make_decoder(seed, stress=stress).double()
    torch.manual_seed(17)
    z = torch.randn(1, 4, latent, latent)
    ref = dec64(z.double())
    y32 = dec(z)
    # conv modules call F.conv2d through torch.nn.functional: patch Conv2d._conv_forward's view of it
    import torch.nn.modules.conv as convmod
    with SplitArithmetic(terms) as sa:
        old = convmod.F.conv2d
        convmod.F.conv2d = sa.conv2d
        try:
            y3 = dec(z)
        finally:
            convmod.F.conv2d = old
    return {"fp32_vs_fp64": rel(y32, ref), f"bf16x{terms}_vs_fp64": rel(y3, ref), f"bf16x{terms}_vs_fp32": rel(y3, y32),
            "out_absmax": float(ref.abs().max()), "info": getattr(dec, "stress_info", None)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--logit-std", type=float, default=8.0)
    ap.add_argument("--terms", type=int, default=3)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--default-init", action="store_true")
    a = ap.parse_args()
    print(run(a.latent, False if a.default_init else a.logit_std, a.terms, a.seed))
