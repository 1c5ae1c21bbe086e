"""Whole-decoder parity on TRAINED-LIKE statistics (hostsim/ldm_decoder.py: apply_stress -- conv gains 1..4 with N(0,1) biases,
zero-sum 3x3 filters on a third of the output channels, GroupNorm gamma 0.2..3 / beta -2..2, 2 % of the residual stream's channels
x100 at mid.block_1, attention logits calibrated to std 8 or 16).  Every other whole-decoder check runs on default-init weights
(activations ~ N(0,1), near-uniform softmax rows, no cancellation); the split-bf16 ("bf16x3") default precision carries 16
significand bits per factor and its error grows with sum|a.w| / |sum a.w|, so the 1e-3 contract of the path (BASELINE.json) is
pinned here where that ratio is large.  Upstream's own warning about this network's magnitudes: scripts/tilevae.py:21-22, 302-304;
the attention body: tile_utils/attn.py:49-72; frozen-statistics GroupNorm: scripts/tilevae.py:218-245.

Checkers: the CPU oracle (tile 64, 96x96 latent) and the same oracle code on torch fp32 on the GPU (tile 256, BASELINE cfg3).
Bars: default precision <= 2e-4 of the output range (5x under the contract), strict-fp32 engine <= 5e-5."""
import pytest
import torch

from hostsim import ldm_decoder as ld
from oracle import gpu_reference as gr
from oracle import vae_oracle as vo

pytestmark = pytest.mark.gpu

TOL_DEFAULT, TOL_F32 = 2e-4, 5e-5


@pytest.fixture(autouse=True)
def _reference_arithmetic():
    nt = torch.get_num_threads()
    torch.set_num_threads(min(32, nt))
    with gr.reference_arithmetic(chunked_attention=False):
        yield
    torch.set_num_threads(nt)


def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _both_precisions(E, hook, z):
    out = hook(z).float().cpu()
    try:
        E.set_precision(E.PRECISION_F32)
        out32 = hook(z).float().cpu()
    finally:
        E.set_precision(E.PRECISION_BF16X3)
    return out, out32


@pytest.mark.parametrize("fast,logit_std", [(True, 8), (True, 16), (False, 8)], ids=["fast-std8", "fast-std16", "slow-std8"])
def test_stress_decode_at_upstream_cpu_default_vs_cpu_oracle(plugin, cuda, fast, logit_std):
    """Full-width stress decoder, 96x96 latent, decoder tile 64 (4 tiles of 86x86, T = 7 396) against the CPU oracle:
    default precision and the strict-fp32 engine."""
    E = plugin.engine
    dec_cpu = ld.make_decoder(7, stress=logit_std)
    torch.manual_seed(21)
    z = torch.randn(1, 4, 96, 96)
    ref = vo.tiled_forward(dec_cpu, z, 64, fast)
    assert torch.isfinite(ref).all()
    dec = ld.make_decoder(7, stress=logit_std).to(cuda)
    dec.original_forward = dec.forward
    hook = plugin.tilevae.VAEHook(dec, 64, is_decoder=True, fast_decoder=fast, fast_encoder=False, color_fix=False)
    out, out32 = _both_precisions(E, hook, z.to(cuda))
    err, err32 = _rel(out, ref), _rel(out32, ref)
    print(f"stress decoder (logit std {logit_std}, fast={fast}) tile 64: bf16x3 vs CPU oracle {err:.2e}, f32 engine vs CPU oracle {err32:.2e}")
    assert err < TOL_DEFAULT, f"stress decode, default precision: rel err {err}"
    assert err32 < TOL_F32, f"stress decode, strict fp32 engine: rel err {err32}"


@pytest.mark.parametrize("fast", [True, False], ids=["fast", "slow"])
def test_stress_assembled_cfg3_decode_vs_oracle_on_gpu(plugin, cuda, fast):
    """BASELINE cfg3's decode (512x512 latent, decoder tile 256 -> the four tile shapes of the 8K decode, T = 77 284-token attention,
    2224^2 convs) of the stress decoder against the oracle on the GPU: default precision and the strict-fp32 engine."""
    E = plugin.engine
    torch.manual_seed(33)
    z = torch.randn(1, 4, 512, 512)
    dec = ld.make_decoder(0, stress=8).to(cuda)
    dec.original_forward = dec.forward
    ref = gr.tiled_forward_gpu(dec, z, 256, fast).cpu()
    assert torch.isfinite(ref).all()
    torch.cuda.empty_cache()
    hook = plugin.tilevae.VAEHook(dec, 256, is_decoder=True, fast_decoder=fast, fast_encoder=False, color_fix=False)
    out, out32 = _both_precisions(E, hook, z.to(cuda))
    assert out.shape == ref.shape == (1, 3, 4096, 4096)
    err, err32 = _rel(out, ref), _rel(out32, ref)
    print(f"stress decoder, assembled cfg3 decode (fast={fast}): bf16x3 vs oracle {err:.2e}, f32 engine vs oracle {err32:.2e}")
    assert err < TOL_DEFAULT, f"stress cfg3 decode (fast={fast}), default precision: rel err {err}"
    assert err32 < TOL_F32, f"stress cfg3 decode (fast={fast}), strict fp32 engine: rel err {err32}"


@pytest.mark.parametrize("logit_std", [8, 32], ids=["std8", "std32"])
def test_stress_attention_alone(plugin, cuda, logit_std):
    """The attention kernel on peaky rows: q / k scaled so the logits have std 8 and 32 (almost one-hot rows), T = 20 000, C = 512,
    split-bf16 and exact kernels vs torch fp64 on the GPU (tile_utils/attn.py:49-72)."""
    E = plugin.engine
    T, C = 20000, 512
    g = torch.Generator(device="cpu").manual_seed(5)
    f = (logit_std ** 0.5)
    q = (torch.randn(1, C, T, generator=g) * f).to(cuda)
    k = (torch.randn(1, C, T, generator=g) * f).to(cuda)
    v = torch.randn(1, C, T, generator=g).to(cuda)
    scale = float(C ** -0.5)
    ref = torch.empty(1, C, T, dtype=torch.float64, device=cuda)
    qt, kd, vd = q.permute(0, 2, 1).double(), k.double(), v.double()
    for i in range(0, T, 2048):
        w_ = torch.softmax(torch.bmm(qt[:, i:i + 2048], kd) * scale, dim=2)
        ref[:, :, i:i + 2048] = torch.bmm(vd, w_.permute(0, 2, 1))
    vt = v.permute(0, 2, 1).contiguous()
    err = _rel(E.vae_attn(q, k, vt, scale).double(), ref)
    err_x = _rel(E.vae_attn(q, k, vt, scale, exact=True).double(), ref)
    print(f"attention at logit std {logit_std}: bf16x3 {err:.2e}, exact {err_x:.2e}")
    assert err < TOL_DEFAULT and err_x < TOL_F32


# ---- the ENCODE direction on the same trained-like statistics (f1's parity rested on default-init weights) -------------------
# upstream: scripts/tilevae.py:155-171 (encoder task queue with the stride-2 Downsample convs), :492-496 (color_fix: statistics frozen only
# up to the first downsample, the rest pooled across tiles), :507-656 (the tile sweep with is_decoder=False, pad 32)
@pytest.mark.parametrize("fast,color_fix,logit_std", [(True, False, 8), (True, False, 16), (False, False, 8), (True, True, 8)],
                         ids=["fast-std8", "fast-std16", "slow-std8", "color_fix-std8"])
def test_stress_encode_vs_cpu_oracle(plugin, cuda, fast, color_fix, logit_std):
    """Full-width stress ENCODER, 168 x 136 image, encoder tile 64 (pad 32: 3 x 3 tiles of <= 128 px) against the CPU oracle: fast mode, slow mode
    (every norm pooled -- incl. the narrow conv_out, cout 8, behind the pooled norm_out on the record path) and color_fix, default precision and
    the strict-fp32 engine."""
    E = plugin.engine
    enc_cpu = ld.make_encoder(7, stress=logit_std)
    torch.manual_seed(23)
    x = torch.randn(1, 3, 168, 136)
    ref = vo.tiled_forward(enc_cpu, x, 64, fast, is_decoder=False, color_fix=color_fix)
    assert torch.isfinite(ref).all() and ref.shape == (1, 8, 21, 17)
    enc = ld.make_encoder(7, stress=logit_std).to(cuda)
    enc.original_forward = enc.forward
    hook = plugin.tilevae.VAEHook(enc, 64, is_decoder=False, fast_decoder=False, fast_encoder=fast, color_fix=color_fix)
    out, out32 = _both_precisions(E, hook, x.to(cuda))
    err, err32 = _rel(out, ref), _rel(out32, ref)
    l2 = ((out - ref).double().norm() / ref.double().norm()).item()
    print(f"stress encoder (logit std {logit_std}, fast={fast}, color_fix={color_fix}) tile 64: bf16x3 vs CPU oracle {err:.2e} (rel L2 {l2:.2e}), f32 engine vs CPU oracle {err32:.2e}")
    assert err < TOL_DEFAULT, f"stress encode, default precision: rel err {err}"
    assert err32 < TOL_F32, f"stress encode, strict fp32 engine: rel err {err32}"


def test_stress_encode_tile_at_upstream_recommended_size_vs_oracle_on_gpu(plugin, cuda):
    """One 3072-class tile of the stress encoder (a 6144 x 6144 image at encoder tile 3072 = 2 x 2 tiles of 3104^2 px, T = 150 544-token attention,
    fast mode) against the oracle's encode of the same tile on the GPU: default precision and the strict-fp32 engine."""
    E = plugin.engine
    enc = ld.make_encoder(0, stress=8).to(cuda)
    enc.original_forward = enc.forward
    x = torch.randn(1, 3, 6144, 6144, generator=torch.Generator().manual_seed(1)).to(cuda)
    (ob, crop), = gr.tiled_forward_gpu(enc, x, 3072, True, is_decoder=False, only_tiles=[2])
    crop = crop.cpu()
    assert torch.isfinite(crop).all()
    torch.cuda.empty_cache()
    hook = plugin.tilevae.VAEHook(enc, 3072, is_decoder=False, fast_decoder=False, fast_encoder=True, color_fix=False)
    out, out32 = _both_precisions(E, hook, x)
    den = out32.abs().max().item()
    cut = lambda t: t[:, :, ob[2]:ob[3], ob[0]:ob[1]]      # noqa: E731
    err = (cut(out) - crop).abs().max().item() / den
    err32 = (cut(out32) - crop).abs().max().item() / den
    print(f"stress encoder, 6144^2 at encoder tile 3072, tile 2 vs the oracle on the GPU: bf16x3 {err:.2e}, f32 engine {err32:.2e}")
    assert err < TOL_DEFAULT and err32 < TOL_F32
