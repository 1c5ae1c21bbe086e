"""Numerical model of the split-bf16 ("bf16x3") arithmetic the MFMA kernels use (csrc/vae_conv_bf16x3.hip,
vae_conv1x1_bf16x3.hip, vae_attn_bf16x3.hip) -- an executable form of DESIGN.md's accuracy claim, no GPU needed.

Every fp32 operand is split x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (round to nearest even, as v_cvt_pk_bf16_f32),
and a product is formed as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  with fp32 accumulation.  Claims checked here:
  * hi + lo carries >= 16 significand bits of x: |x - (hi + lo)| <= 2^-16 |x|  (up to an ulp of slack);
  * one product is within ~3 * 2^-16 of the exact product (the dropped a_lo*b_lo term and the two split residues);
  * a K = 4608 dot product (3x3 conv, cin = 512) is within 1e-5 of its fp64 value relative to sum |a_k b_k|, and the
    plain-bf16 (1 MFMA) alternative is ~2 orders of magnitude worse -- why the engine pays for three MFMAs."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hostsim import ldm_decoder as ld  # noqa: E402


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round to nearest even) -> fp32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split(x):
    hi = bf16_round(x)
    lo = bf16_round((x.astype(np.float32) - hi).astype(np.float32))
    return hi, lo


def test_split_keeps_sixteen_bits():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-20, 20, 200000))).astype(np.float32)
    hi, lo = split(x)
    rel = np.abs(x.astype(np.float64) - (hi.astype(np.float64) + lo.astype(np.float64))) / np.abs(x.astype(np.float64))
    assert rel.max() <= 2.0 ** -16 * 1.01


def test_product_error_bound():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(200000).astype(np.float32)
    b = rng.standard_normal(200000).astype(np.float32)
    ah, al = split(a)
    bh, bl = split(b)
    p = (al.astype(np.float64) * bh + ah.astype(np.float64) * bl + ah.astype(np.float64) * bh)
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(p - exact) / np.abs(exact)
    assert rel.max() <= 3.0 * 2.0 ** -16


def test_dot_product_accuracy_vs_plain_bf16():
    rng = np.random.default_rng(2)
    K, M = 4608, 256
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = rng.standard_normal((M, K)).astype(np.float32)
    ah, al = split(a)
    bh, bl = split(b)
    # fp32 accumulation of the three-term products in K order (MFMA accumulators are fp32)
    acc = np.zeros(M, dtype=np.float32)
    for k in range(0, K, 16):   # one 16-deep MFMA k-step at a time
        for t in ((al, bh), (ah, bl), (ah, bh)):
            acc = (acc + np.sum(t[0][:, k:k + 16].astype(np.float64) * t[1][:, k:k + 16], axis=1)).astype(np.float32)
    exact = np.sum(a.astype(np.float64) * b, axis=1)
    scale = np.sum(np.abs(a.astype(np.float64) * b), axis=1)
    err3 = np.abs(acc - exact) / scale
    plain = np.sum(ah.astype(np.float64) * bh, axis=1)
    err1 = np.abs(plain - exact) / scale
    assert err3.max() < 1e-5
    assert np.median(err1) > 30 * np.median(err3)


def _dot3(a, b, terms=3):
    """K-ordered fp32 accumulation of the split products, one 16-deep MFMA k-step at a time (lo*hi, hi*lo, [lo*lo], hi*hi)."""
    ah, al = split(a)
    bh, bl = split(b)
    seq = [(al, bh), (ah, bl)] + ([(al, bl)] if terms == 4 else []) + [(ah, bh)]
    acc = np.zeros(a.shape[0], dtype=np.float32)
    for k in range(0, a.shape[1], 16):
        for t in seq:
            acc = (acc + np.sum(t[0][:, k:k + 16].astype(np.float64) * t[1][:, k:k + 16], axis=1)).astype(np.float32)
    return acc


@pytest.mark.parametrize("ratio", [10, 100, 1000])
def test_dot_product_under_cancellation(ratio):
    """The bound stated against |sum a_k b_k| itself (not sum |a_k b_k|, which hides cancellation by construction): operands with
    mean >> spread against zero-sum filters, tuned so that  R = sum|a_k b_k| / |sum a_k b_k|  is `ratio`.  The split-bf16 dot
    product is within  R * 2^-16 / sqrt(K/16)  ... measured: <= 2.5e-7 * R relative to the exact result at K = 4608, i.e. 2.5e-4
    at R = 1000 -- the regime where a layer needs the exact kernel; a plain fp32 dot product (the oracle's arithmetic) is itself
    only good to ~1e-7 * R there."""
    rng = np.random.default_rng(ratio)
    K, M = 4608, 512
    # activations: post-SiLU like (positive mean, smaller spread); weights: zero-sum part + a small common part that sets R
    a = (1.0 + 0.3 * rng.standard_normal((M, K))).astype(np.float32)
    w0 = rng.standard_normal((M, K))
    w0 -= w0.mean(axis=1, keepdims=True)
    b = w0.astype(np.float32)
    exact0 = np.sum(a.astype(np.float64) * b, axis=1)
    scale = np.sum(np.abs(a.astype(np.float64) * b), axis=1)
    # shift every row's weights by a constant so that |sum| = scale / ratio
    shift = (scale / ratio - exact0) / np.sum(a.astype(np.float64), axis=1)
    b = (w0 + shift[:, None]).astype(np.float32)
    exact = np.sum(a.astype(np.float64) * b, axis=1)
    scale = np.sum(np.abs(a.astype(np.float64) * b), axis=1)
    R = scale / np.abs(exact)
    assert 0.5 * ratio < np.median(R) < 2.0 * ratio
    err3 = np.abs(_dot3(a, b) - exact) / np.abs(exact)
    err4 = np.abs(_dot3(a, b, terms=4) - exact) / np.abs(exact)
    f32 = np.zeros(M, dtype=np.float32)
    for k in range(K):
        f32 = (f32 + a[:, k] * b[:, k]).astype(np.float32)
    err32 = np.abs(f32 - exact) / np.abs(exact)
    print(f"R={ratio}: bf16x3 max {err3.max():.2e} median {np.median(err3):.2e}; +lo*lo max {err4.max():.2e}; fp32 chain max {err32.max():.2e}")
    assert err3.max() <= 2.5e-7 * ratio * 2.0
    assert np.median(err3) <= 1.0e-7 * ratio
    assert err4.max() <= err3.max() * 1.5      # the 4th term is not where the error is: the 2^-17 split residues are


def test_stress_recipe_is_what_it_says():
    """The committed recipe: zero-sum filters, massive channels in the residual stream, calibrated logits (CPU arithmetic only)."""
    dec = ld.make_decoder(7, stress=16)
    info = dec.stress_info
    assert abs(info["logit_std"] - 16.0) < 0.05 and len(info["massive_channels"]) == 10
    w = dec.mid.block_2.conv1.weight
    assert w[::3].sum(dim=(1, 2, 3)).abs().max() < 1e-3 * w[::3].abs().sum(dim=(1, 2, 3)).min()
    g = dec.up[1].block[0].norm1
    assert 0.2 <= float(g.weight.min()) and float(g.weight.max()) <= 3.0 and -2.0 <= float(g.bias.min()) and float(g.bias.max()) <= 2.0
    with torch.no_grad():
        h = dec.mid.block_1(dec.conv_in(torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(0))))
    per_ch = h.abs().amax(dim=(0, 2, 3))
    assert per_ch[info["massive_channels"]].min() > 10 * per_ch.median()




@pytest.mark.parametrize("logit_std", [8, 16])
def test_whole_decoder_model_on_stress_statistics(logit_std):
    """tools/bf16x3_model.py: the FULL-WIDTH stress decoder (eager, 16x16 latent) with every conv and both attention contractions
    in split-bf16 arithmetic against the same network in fp64 -- the CPU-side prediction of tests/test_gpu_vae_stress.py:
    <= 2e-4 of the output range (observed 2-3e-5: GroupNorm re-normalises every conv input, so the massive channels and zero-sum
    filters do not compound)."""
    import bf16x3_model as bm
    r = bm.run(16, logit_std, 3, seed=7)
    print(r)
    assert r["bf16x3_vs_fp64"] < 2e-4
    assert r["fp32_vs_fp64"] < 2e-5
