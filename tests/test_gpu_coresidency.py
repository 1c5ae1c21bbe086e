"""conv_in next to other kernels on the same CU (round 6).  Upstream runs the decoder's conv_in as an nn.Conv2d task like any other
(scripts/tilevae.py:115-137); the fp32 kernel that serves it here (csrc/vae_conv.hip: k_conv3x3_fewcin) returned wrong lanes in its round-5
form whenever its waves shared a CU with another kernel's MFMA waves -- another stream of the same process is enough (DESIGN.md 3.5,
profiles/r6j).  Every other GPU test runs its kernels alone on the chip, so this one overlaps conv_in with hand-over 3x3 convs on a side
stream (the neighbour that produced wrong values in 100 of 100 launches) and asks for the bit pattern of the launch that ran alone.
The blend (bit-exact by contract, packed-fp32 `op_sel` encodings in its MoD path) and a whole tiled decode (every kernel of the VAE path,
fast and slow mode, decoder and encoder) get the same treatment."""
import threading

import pytest
import torch

from hostsim import ldm_decoder as ld

pytestmark = pytest.mark.gpu


def _coef(B, C, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return torch.cat([torch.rand(B, 1, C, generator=g) * 1.5 + 0.25, torch.randn(B, 1, C, generator=g) * 0.5], dim=1).contiguous().to(dev)


def _neighbours(E, dev):
    """Two hand-over convs (norm + SiLU applied while staging, split-bf16 MFMAs): ~1 ms each, two blocks per CU with LDS to spare."""
    torch.manual_seed(3)
    c512 = torch.nn.Conv2d(512, 512, 3, padding=1).to(dev)
    c256 = torch.nn.Conv2d(256, 256, 3, padding=1).to(dev)
    p512 = E.PackedConv(c512.weight.detach(), c512.bias.detach())
    p256 = E.PackedConv(c256.weight.detach(), c256.bias.detach())
    x512, x256 = torch.randn(2, 512, 278, 278, device=dev), torch.randn(1, 256, 556, 556, device=dev)
    k512, k256 = _coef(2, 512, 1, dev), _coef(1, 256, 2, dev)
    return lambda: (p512(x512, pre_gn=k512), p256(x256, pre_gn=k256))


@pytest.mark.parametrize("cin,cout,hw", [(4, 512, 278), (3, 128, 600)])
def test_conv_in_is_bit_stable_next_to_mfma_kernels(plugin, cuda, cin, cout, hw):
    E, dev = plugin.engine, cuda
    torch.manual_seed(0)
    c = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    pc = E.PackedConv(c.weight.detach(), c.bias.detach())
    z = torch.randn(2, cin, hw, hw, device=dev)
    alone = pc(z).clone()
    torch.cuda.synchronize()
    with torch.no_grad():
        want = torch.nn.functional.conv2d(z, c.weight, c.bias, padding=1)
    assert (alone - want).abs().max().item() < 1e-5
    busy = _neighbours(E, dev)
    side = torch.cuda.Stream()
    bad = 0
    for _ in range(40):
        with torch.cuda.stream(side):
            busy()
        y = pc(z)
        side.synchronize()
        bad += int(not torch.equal(y, alone))
    print(f"conv_in {cin}->{cout} {hw}^2 overlapped with hand-over convs on a side stream: {bad} of 40 launches differ from the launch that ran alone")
    assert bad == 0


@pytest.mark.parametrize("method", ["md", "mod"])
def test_blend_is_bit_stable_next_to_mfma_kernels(plugin, cuda, method):
    import bench
    E, dev = plugin.engine, cuda
    plan, gather, blend, nbytes = bench.blend_setup(E, dev, 512, 512, 96, 48, 8, method)
    blend()
    torch.cuda.synchronize()
    alone = blend.out.clone()
    busy = _neighbours(E, dev)
    side = torch.cuda.Stream()
    bad = 0
    for r in range(200):
        if r % 20 == 0:
            with torch.cuda.stream(side):
                busy()
        blend.out.fill_(float("nan"))
        blend()
        bad += int(not torch.equal(blend.out, alone))
    side.synchronize()
    print(f"blend ({method}, 96 / 48 grid) overlapped with hand-over convs: {bad} of 200 launches differ")
    assert bad == 0


@pytest.mark.parametrize("what", ["decode-fast", "decode-slow", "encode-fast"])
def test_tiled_vae_is_bit_stable_next_to_mfma_kernels(plugin, cuda, what):
    """vae_tile_forward (scripts/tilevae.py:577-737 upstream) of a 160x160 latent at decoder tile 64 (9 tiles, T = 7 396) / a 1280^2 image
    at encoder tile 512, alone and with another THREAD of the process launching hand-over convs on its own stream the whole time: the
    two results must be the same bits (the decode is deterministic: same tiles, same stacking, same kernels)."""
    E, dev = plugin.engine, cuda
    dec = what.startswith("decode")
    fast = what.endswith("fast")
    net = (ld.make_decoder(5) if dec else ld.make_encoder(5)).to(dev)
    net.original_forward = net.forward
    hook = plugin.tilevae.VAEHook(net, 64 if dec else 512, is_decoder=dec, fast_decoder=fast, fast_encoder=fast, color_fix=False)
    torch.manual_seed(17)
    x = (torch.randn(1, 4, 160, 160) if dec else torch.randn(1, 3, 1280, 1280)).to(dev)
    alone = hook(x).clone()
    torch.cuda.synchronize()
    busy = _neighbours(E, dev)
    stop = threading.Event()

    def load():
        torch.cuda.set_device(dev)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            while not stop.is_set():
                busy()
                side.synchronize()

    th = threading.Thread(target=load)
    th.start()
    try:
        bad = 0
        for _ in range(3):
            y = hook(x)
            torch.cuda.synchronize()
            bad += int(not torch.equal(y, alone))
    finally:
        stop.set()
        th.join()
    print(f"{what}: {bad} of 3 runs next to hand-over convs on another stream differ from the run that had the chip alone")
    assert bad == 0
