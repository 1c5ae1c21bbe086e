"""
HOST STAND-IN (test / bench infrastructure; hostsim/) -- restatement of the third-party VAE network the reference walks.

The network itself is NOT under /root/reference: it lives in `ldm.modules.diffusionmodules.model`
(Stability-AI/stablediffusion; SDXL: `sgm.modules.diffusionmodules.model`).  The reference pins no version
(no manifest); A1111 pins stablediffusion@cf1d67a6 (recollection, unverifiable offline).  What is restated here
is the published architecture of `Decoder` / `Encoder` / `ResnetBlock` / `AttnBlock` / `Upsample` / `Downsample`
for the SD1.x / SD2.x / SDXL KL-f8 auto-encoder (ch=128, ch_mult=(1,2,4,4), num_res_blocks=2, z_channels=4,
GroupNorm(32, eps=1e-6, affine), nearest-2x upsample followed by a 3x3 conv), anchored on the reference's own call
sites: the attribute walk in scripts/tilevae.py:107-195 and the attention body tile_utils/attn.py:49-72.
Random weights only (no checkpoints offline).

Used by tests/, smoke(), bench.py and the probes as the module whose forward the Tiled-VAE hook replaces and whose (random) weights both
the product and the oracle read; its own eager `forward` is what oracle/vae_oracle.py replays op by op.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def Normalize(c: int) -> nn.GroupNorm:
    return nn.GroupNorm(num_groups=32, num_channels=c, eps=1e-6, affine=True)


class ResnetBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_conv_shortcut = False
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.in_channels != self.out_channels:
            x = self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.in_channels = c
        self.norm = Normalize(c)
        self.q = nn.Conv2d(c, c, 1)
        self.k = nn.Conv2d(c, c, 1)
        self.v = nn.Conv2d(c, c, 1)
        self.proj_out = nn.Conv2d(c, c, 1)

    def forward(self, x):
        h = self.norm(x)
        b, c, hh, ww = h.shape
        q = self.q(h).reshape(b, c, hh * ww).permute(0, 2, 1)
        k = self.k(h).reshape(b, c, hh * ww)
        v = self.v(h).reshape(b, c, hh * ww)
        w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
        h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
        return x + self.proj_out(h)


class Upsample(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.with_conv = True
        self.conv = nn.Conv2d(c, c, 3, 1, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Downsample(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.with_conv = True
        self.conv = nn.Conv2d(c, c, 3, 2, 0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class Decoder(nn.Module):
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4,
                 give_pre_end=False, tanh_out=False):
        super().__init__()
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            up = nn.Module()
            up.block = block
            up.attn = nn.ModuleList()
            if i_level != 0:
                up.upsample = Upsample(block_in)
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for i_level in reversed(range(self.num_resolutions)):
            for blk in self.up[i_level].block:
                h = blk(h)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        if self.give_pre_end:
            return h
        h = self.conv_out(F.silu(self.norm_out(h)))
        return torch.tanh(h) if self.tanh_out else h


class Encoder(nn.Module):
    def __init__(self, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4, double_z=True):
        super().__init__()
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            down = nn.Module()
            down.block = block
            down.attn = nn.ModuleList()
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in)
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, 1, 1)

    def forward(self, x):
        h = self.conv_in(x)
        for i_level in range(self.num_resolutions):
            for blk in self.down[i_level].block:
                h = blk(h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.conv_out(F.silu(self.norm_out(h)))


def _logit_std(attn: "AttnBlock", h: torch.Tensor) -> float:
    """Standard deviation of the scaled attention logits q.k / sqrt(C) of `attn` on the (already normalised) input h."""
    b, c, hh, ww = h.shape
    q = attn.q(h).reshape(b, c, hh * ww).permute(0, 2, 1)
    k = attn.k(h).reshape(b, c, hh * ww)
    return float((torch.bmm(q, k) * (int(c) ** -0.5)).std())


@torch.no_grad()
def apply_stress(net: nn.Module, seed: int = 0, logit_std: float = 8.0, massive_frac: float = 0.02,
                 massive_gain: float = 100.0) -> dict:
    """Trained-like ("stress") statistics for a random-init Decoder / Encoder -- the committed recipe the parity tests and
    bench.py's `parity.rel_err_vs_oracle_stress` leg use.  Default nn.init gives activations ~ N(0,1), near-uniform softmax rows
    (logit std ~ 1.5) and no cancellation; the checkpoints the configs name do not look like that (upstream warns of fp16 overflow
    in its own VAE path: scripts/tilevae.py:21-22, 302-304).  In module order, from one generator:
      * every conv: weight x g, g log-uniform in [1, 4]; bias ~ N(0, 1);
      * every 3x3 conv: each third output channel made ZERO-SUM over (cin, ky, kx) -- its result is the difference of large
        partial sums whenever the (post-SiLU, mostly positive) input has mean >> std;
      * every GroupNorm: gamma uniform in [0.2, 3], beta uniform in [-2, 2];
      * mid.block_1.conv2 (writes the residual stream): `massive_frac` of its output channels x `massive_gain` (weights and
        bias) -- the few huge channels trained auto-encoders carry;
      * mid.attn_1.q / .k: rescaled so that the scaled logits of a fixed seeded 16x16 latent have std = `logit_std`
        (8: peaky rows; 16: almost one-hot).
    Returns what was done (for the bench line / test messages)."""
    g = torch.Generator().manual_seed(1000 + seed)
    n_zero_sum = 0
    for name, m in net.named_modules():
        if isinstance(m, nn.Conv2d):
            gain = float(torch.exp(torch.rand((), generator=g) * math.log(4.0)))
            m.weight.mul_(gain)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g))
            if m.kernel_size == (3, 3) and m.out_channels >= 3:
                m.weight[::3] -= m.weight[::3].mean(dim=(1, 2, 3), keepdim=True)
                n_zero_sum += len(range(0, m.out_channels, 3))
        elif isinstance(m, nn.GroupNorm):
            m.weight.copy_(0.2 + 2.8 * torch.rand(m.weight.shape, generator=g))
            m.bias.copy_(-2.0 + 4.0 * torch.rand(m.bias.shape, generator=g))
    info = {"recipe": "stress", "seed": seed, "zero_sum_filters": n_zero_sum}
    mid = getattr(net, "mid", None)
    if mid is not None:
        c2 = mid.block_1.conv2
        n_massive = max(1, int(round(massive_frac * c2.out_channels)))
        idx = torch.randperm(c2.out_channels, generator=g)[:n_massive]
        c2.weight[idx] *= massive_gain
        c2.bias[idx] *= massive_gain
        info["massive_channels"] = sorted(int(i) for i in idx)
        # calibrate the logits on the network's own activations at the attention (fixed seeded probe input)
        zc = net.conv_in.in_channels
        probe = torch.randn(1, zc, 16, 16, generator=g)
        h = net.conv_in(probe)
        if isinstance(net, Encoder):
            for lvl in range(net.num_resolutions):
                for blk in net.down[lvl].block:
                    h = blk(h)
                if lvl != net.num_resolutions - 1:
                    h = net.down[lvl].downsample(h)
        h = mid.attn_1.norm(mid.block_1(h))
        s0 = _logit_std(mid.attn_1, h)
        f = math.sqrt(logit_std / max(s0, 1e-12))
        for conv in (mid.attn_1.q, mid.attn_1.k):
            conv.weight.mul_(f)
            conv.bias.mul_(f)
        info["logit_std_before"] = round(s0, 3)
        info["logit_std"] = round(_logit_std(mid.attn_1, h), 3)
    return info


def make_decoder(seed: int = 0, small: bool = False, stress=False, **kw) -> Decoder:
    """Random-weight decoder.  `small=True` gives a CPU-cheap net with the SAME topology (4 levels, attention in the
    middle, 32-group norms) at ch=32 so that oracle parity tests finish in seconds.  `stress` = True / a logit std
    (8, 16, ...) applies `apply_stress` (trained-like statistics) on top of the default init; the applied recipe is left in
    `dec.stress_info`."""
    torch.manual_seed(seed)
    if small:
        kw.setdefault("ch", 32)
    dec = Decoder(**kw).eval()
    if stress:
        dec.stress_info = apply_stress(dec, seed, logit_std=8.0 if stress is True else float(stress))
        for p in dec.parameters():
            p.requires_grad_(False)
        return dec
    # default nn.init leaves GroupNorm affine at (1, 0); perturb so a gamma/beta mix-up cannot hide
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in dec.modules():
            if isinstance(m, nn.GroupNorm):
                m.weight.add_(0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.add_(0.1 * torch.randn(m.bias.shape, generator=g))
    for p in dec.parameters():
        p.requires_grad_(False)
    return dec


def make_encoder(seed: int = 0, small: bool = False, stress=False, **kw) -> Encoder:
    torch.manual_seed(seed)
    if small:
        kw.setdefault("ch", 32)
    enc = Encoder(**kw).eval()
    if stress:
        enc.stress_info = apply_stress(enc, seed, logit_std=8.0 if stress is True else float(stress))
    for p in enc.parameters():
        p.requires_grad_(False)
    return enc
