"""Numerical model of the split-bf16 ("bf16x3") arithmetic the MFMA kernels use (csrc/vae_conv_bf16x3.hip,
vae_conv1x1_bf16x3.hip, vae_attn_bf16x3.hip) -- an executable form of DESIGN.md's accuracy claim, no GPU needed.

Every fp32 operand is split x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (round to nearest even, as v_cvt_pk_bf16_f32),
and a product is formed as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  with fp32 accumulation.  Claims checked here:
  * hi + lo carries >= 16 significand bits of x: |x - (hi + lo)| <= 2^-16 |x|  (up to an ulp of slack);
  * one product is within ~3 * 2^-16 of the exact product (the dropped a_lo*b_lo term and the two split residues);
  * a K = 4608 dot product (3x3 conv, cin = 512) is within 1e-5 of its fp64 value relative to sum |a_k b_k|, and the
    plain-bf16 (1 MFMA) alternative is ~2 orders of magnitude worse -- why the engine pays for three MFMAs."""
import numpy as np


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
