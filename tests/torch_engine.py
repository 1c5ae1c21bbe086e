"""Torch doubles of the engine surface the HOST logic calls (test infrastructure).

The product's VAEHook / seqpar executor talk to the `mdtile` module (HIP, GPU only).  The CPU tests of that host logic --
tile scheduling, fast / slow / semi-fast GroupNorm handling, crop + assemble, tile sharding and the sequence-parallel
estimator across ranks -- inject these plain-torch stand-ins with the same call signatures, so the logic is checked
against the oracle without a GPU.  Nothing here is imported by the product."""
import math

import torch
import torch.nn.functional as F

from oracle import vae_oracle as vo


class TorchConv:
    """Stand-in for mdtile.PackedConv around an nn.Conv2d."""

    def __init__(self, conv):
        self.conv = conv
        self.ksize = conv.kernel_size[0]
        self.cin, self.cout = conv.in_channels, conv.out_channels
        # ldm Downsample.conv has stride 2 / padding 0; everything else is 'same' stride 1
        self.down = conv.stride == (2, 2)

    def fuses_pre_gn(self, upsample2x=False, token_major=False, exact=False):
        return self.ksize == 3 and not self.down and not upsample2x and not token_major and self.cin % 16 == 0 and self.cout >= 32

    stats_left = 0       # calls that also left the statistics of their output (mdtile_conv2d_gn_stats / _rec_stats stand-ins)

    def leaves_stats(self, groups=32, upsample2x=False, rec=False):
        """PackedConv.leaves_stats: the stand-in says yes wherever the product's 128-cout kernels could (whole quads per group), scaled to the
        small test decoders: couts in whole groups of >= 1."""
        return self.ksize == 3 and not self.down and self.cout % groups == 0 and (rec or not upsample2x)

    def __call__(self, x, residual=None, upsample2x=False, token_major=False, exact=False, pre_gn=None):
        if pre_gn is not None:
            x = F.silu(x * pre_gn[:, 0, :, None, None] + pre_gn[:, 1, :, None, None])
        if upsample2x:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        y = self.conv(x)
        if residual is not None:
            y = y + residual
        if token_major:
            B, C, H, W = y.shape
            y = y.permute(0, 2, 3, 1).reshape(B, H * W, C).contiguous()
        return y

    def call_stats(self, x, pre_gn, residual=None, groups=32):
        """PackedConv.call_stats: always (y, (var, mean))."""
        assert pre_gn is not None
        y = self(x, residual=residual, pre_gn=pre_gn)
        TorchConv.stats_left += 1
        return y, vo.get_var_mean(y, groups)

    def down2(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class TorchEngine:
    """Stand-in for the `mdtile` module as scripts/tilevae.py uses it."""

    class MdtileError(RuntimeError):
        pass

    def require_device(self, dev):
        return None

    def vae_split_tiles(self, h, w, tile_size, is_decoder):
        return vo.split_tiles(h, w, tile_size, is_decoder)

    def vae_fast_input(self, z, tile_size):
        return vo.fast_mode_input(z, tile_size)

    def vae_best_tile_size(self, lowerbound, upperbound):
        return vo.best_tile_size(lowerbound, upperbound)

    def tanh(self, x, out=None):
        return torch.tanh(x)

    def gather_rect(self, z, x, y, w, h):
        return z[:, :, y:y + h, x:x + w].clone()

    def crop_store(self, tile, in_bbox, out_bbox, result, is_decoder=True):
        result[:, :, out_bbox[2]:out_bbox[3], out_bbox[0]:out_bbox[1]] = vo.crop_valid_region(tile, in_bbox, out_bbox, is_decoder)

    def gn_stats(self, x, groups=32):
        return vo.get_var_mean(x, groups)

    def gn_pool(self, means, vars_, pixels):
        return vo.pool_stats(list(vars_), list(means), pixels)

    def gn_coeffs(self, mean, var, gamma, beta, C, groups=32, eps=1e-6):
        B = mean.numel() // groups
        cpg = C // groups
        rstd = 1.0 / torch.sqrt(var.view(B, groups, 1) + eps)
        g = gamma.view(1, groups, cpg) if gamma is not None else torch.ones(1, groups, cpg)
        b = beta.view(1, groups, cpg) if beta is not None else torch.zeros(1, groups, cpg)
        a = (rstd * g).reshape(B, C)
        s = (b - mean.view(B, groups, 1) * a.view(B, groups, cpg)).reshape(B, C)
        return torch.stack([a, s], dim=1)

    def gn_apply(self, x, mean, var, gamma, beta, groups=32, eps=1e-6, silu=False, out=None):
        y = vo.custom_group_norm(x, groups, mean, var, gamma, beta, eps)
        y = F.silu(y) if silu else y
        if out is not None:
            out.copy_(y)
            return out
        return y

    def vae_attn(self, q, k, v_tok, scale):
        w = torch.softmax(torch.bmm(q.permute(0, 2, 1), k) * scale, dim=2)
        return torch.bmm(w, v_tok).permute(0, 2, 1).contiguous()


class TorchSeqParOps:
    """Stand-in for mdtile.seqpar.EngineOps (the ops interface of estimate_group_norm_sp)."""

    def ksize(self, conv):
        return conv.ksize

    def fuses_pre_gn(self, conv, upsample):
        return conv.fuses_pre_gn(upsample2x=upsample)

    def conv(self, conv, x, residual=None, upsample2x=False, pre_gn=None, token_major=False):
        return conv(x, residual=residual, upsample2x=upsample2x, token_major=token_major, pre_gn=pre_gn)

    def gn_sums(self, x, row_lo, row_hi):
        B = x.shape[0]
        v = x[:, :, row_lo:row_hi, :].double().reshape(B * 32, -1)
        return torch.stack([v.sum(1), (v * v).sum(1)], dim=1)

    def gn_from_sums(self, sums, count):
        m = sums[:, 0] / count
        v = (sums[:, 1] / count - m * m).clamp_min(0.0)
        return v.float(), m.float()

    def gn_coeffs(self, mean, var, gamma, beta, C):
        return TorchEngine().gn_coeffs(mean, var, gamma, beta, C)

    def gn_apply(self, x, mean, var, gamma, beta, silu, inplace):
        return TorchEngine().gn_apply(x, mean, var, gamma, beta, 32, 1e-6, silu)

    def attn_qk(self, q, k, v_tok, scale):
        return TorchEngine().vae_attn(q, k, v_tok, scale)

    def tanh(self, x):
        return torch.tanh(x)


# ---- record-path doubles: the fast-mode sweep of scripts/tilevae.py (_run_tile_rec) with the hand-over between the 3x3 convs emulated
# in fp32 torch, so that the host logic of that sweep -- which producer applies which norm, the live-window narrowing of the tiles
# (live_windows), stacking by shape -- is checked against the oracle on CPU.
class TorchRec:
    """Stand-in for mdtile.RecImage: the (already activated) fp32 tensor itself."""

    def __init__(self, t):
        self.t = t
        self.shape = tuple(t.shape)

    def batch_slice(self, b0, b1):
        return TorchRec(self.t[b0:b1])

    @staticmethod
    def cat(recs):
        return recs[0] if len(recs) == 1 else TorchRec(torch.cat([r.t for r in recs], dim=0))


class TorchConvRec(TorchConv):
    px_computed = 0      # output pixels x couts of every call_rec (what a narrowed sweep saves)
    window_calls = 0
    mixed_origin_calls = 0     # window calls whose images (stacked tiles) have different origins

    def takes_rec(self, upsample2x=False):
        return self.ksize == 3 and not self.down and self.cin % 32 == 0 and self.cout % 32 == 0

    def call_rec_stats(self, xrec, residual=None, upsample2x=False, family=0, groups=32):
        """PackedConv.call_rec_stats: always (y, (var, mean))."""
        y, _ = self.call_rec(xrec, residual=residual, upsample2x=upsample2x, want_f32=True, want_rec=False)
        TorchConv.stats_left += 1
        return y, vo.get_var_mean(y, groups)

    def call_rec(self, xrec, residual=None, upsample2x=False, want_f32=True, want_rec=False, rec_coef=None, window=None, family=0):
        x = xrec.t
        if window is not None:
            # mdtile_upconv2d_rec_window: the conv of a window of the input whose edges inside the image see the true neighbours ==
            # the same pixels of the whole-image result
            assert upsample2x and residual is None
            y0, x0, h, w = window
            B = x.shape[0]
            y0 = [y0] * B if isinstance(y0, int) else list(y0)       # one origin for all images, or one per image (stacked tiles)
            x0 = [x0] * B if isinstance(x0, int) else list(x0)
            assert len(y0) == len(x0) == B and (B <= 8 or all(y0[b] == y0[b & 7] and x0[b] == x0[b & 7] for b in range(B)))
            assert all(0 <= a and 0 <= c and a + h <= x.shape[2] and c + w <= x.shape[3] for a, c in zip(y0, x0))
            full = self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))
            y = torch.stack([full[b, :, 2 * y0[b]:2 * (y0[b] + h), 2 * x0[b]:2 * (x0[b] + w)] for b in range(B)])
            TorchConvRec.window_calls += 1
            TorchConvRec.mixed_origin_calls += int(len(set(zip(y0, x0))) > 1)
        else:
            if upsample2x:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            y = self.conv(x)
        TorchConvRec.px_computed += y.shape[0] * y.shape[1] * y.shape[2] * y.shape[3]
        if residual is not None:
            y = y + residual
        yr = None
        if want_rec:
            yr = TorchRec(F.silu(y * rec_coef[:, 0, :, None, None] + rec_coef[:, 1, :, None, None]) if rec_coef is not None else y)
        return (y if want_f32 else None), yr


class TorchConvCounting(TorchConv):
    """TorchConv (no record form) that books its output elements in the same counter as TorchConvRec."""

    def __call__(self, x, residual=None, upsample2x=False, token_major=False, exact=False, pre_gn=None):
        out = super().__call__(x, residual, upsample2x, token_major, exact, pre_gn)
        if self.ksize == 3:
            TorchConvRec.px_computed += out.numel()
        return out


class TorchEngineRec(TorchEngine):
    def rec_from_f32(self, x, coef=None):
        return TorchRec(F.silu(x * coef[:, 0, :, None, None] + coef[:, 1, :, None, None]) if coef is not None else x)
