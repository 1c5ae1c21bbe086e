"""
Sequence-parallel (row-band) execution of the Tiled-VAE fast-mode GroupNorm estimator across GPUs.

Why: in fast mode upstream first runs the WHOLE network once on a down-sampled latent to freeze the GroupNorm statistics
(`estimate_group_norm`, scripts/tilevae.py:464-505), then sweeps the tiles.  On one GPU that pass is one tile's worth of
work out of 17 (8K image, decoder tile 256).  With the tiles dealt over 8 GPUs every rank would still repeat the full
estimator: 2 tiles + 1 estimator = 3 units per rank against 17 on one GPU caps strong scaling at 5.7x.  Here the estimator
itself is split: every rank owns a contiguous band of rows of the (single, untiled) estimator activation and the network
runs once across the ranks, exactly (no tile approximation):

  * 3x3 convs need one row from each neighbour: the band tensor carries a 1-row halo slot on every side that faces
    another band ([B, C, ht + rows + hb, W]; no slot at the image border, so the kernels' own zero padding is the image's
    padding).  Halos are refreshed by a neighbour send/recv right before a 3x3 conv consumes them (xGMI is point to
    point: two 0.5-1 MB messages per conv, no collective); the conv's own halo output rows are garbage until the next
    refresh.  Point-wise steps (1x1 convs, GroupNorm apply, SiLU, residual add) run on the extended tensor as is.
  * nearest-2x upsample + conv: the halo rows produce two output rows each; the inner one is exact (it only needs the
    halo row and the first own row), the outer one is dropped -> the result is again a band with VALID 1-row halos.
  * GroupNorm statistics (`get_var_mean`, tilevae.py:207-215): fp64 (sum, sum of squares) over the OWN rows
    (mdtile_gn_sums), all-reduce(sum) of 2 x B x 32 doubles, then mean / biased variance (mdtile_gn_from_sums).
  * attention (tile_utils/attn.py:49-72): queries stay local, keys / values of all bands are all-gathered once
    (2 x 134 MB at 256 x 256 tokens, C = 512) and the flash kernel runs with Tq != Tk (mdtile_vae_attn_qk).

The executor is written against a small `ops` interface so that the host logic (partition, halo protocol, gathers,
statistics) is exercised on CPU by the world-size-2/3 gloo tests with torch ops injected by the TEST
(tests/test_seqpar_gloo.py); the product always passes `EngineOps` (mdtile C-ABI calls, GPU only).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def row_bounds(H: int, world: int) -> List[int]:
    """Band r owns rows [bounds[r], bounds[r+1]) -- sizes differ by at most one."""
    return [H * r // world for r in range(world + 1)]


class BandComm:
    """Neighbour / group communication of one band, on the job's data plane (mdtile/sharding.py comm_*): the engine's own RCCL
    communicator behind the C ABI when the process has one (mdtile_shard_p2p / _allreduce_stats / _allgather on torch's current
    stream), torch.distributed otherwise (gloo in the CPU tests and single-GPU multi-process checks: staged through the host)."""

    def __init__(self, rank: int, world: int, group=None):
        self.rank, self.world, self.group = rank, world, group

    def exchange_halos(self, x: torch.Tensor, ht: int, hb: int, rows: int) -> None:
        """Refresh the halo slots of x [B, C, ht + rows + hb, W] in place: my first own row goes to the upper neighbour's
        bottom slot, my last own row to the lower neighbour's top slot, and vice versa -- both directions in ONE grouped
        exchange (with two ranks both halves meet the same peer; xGMI links are per pair, so up and down overlap)."""
        if self.world == 1 or (ht == 0 and hb == 0):
            return
        from mdtile import sharding
        ops, recvs = [], []
        if ht:
            send = x[:, :, ht:ht + 1, :].contiguous()
            recv = torch.empty_like(send)
            ops.append((self.rank - 1, send, recv))
            recvs.append((0, recv))
        if hb:
            send = x[:, :, ht + rows - 1:ht + rows, :].contiguous()
            recv = torch.empty_like(send)
            ops.append((self.rank + 1, send, recv))
            recvs.append((ht + rows, recv))
        sharding.comm_p2p(ops, self.group)
        for row, recv in recvs:
            x[:, :, row:row + 1, :].copy_(recv)

    def allreduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return t
        from mdtile import sharding
        return sharding.comm_allreduce_sum(t, self.group)

    def allgather_cat(self, t: torch.Tensor, dim: int, sizes: Sequence[int]) -> torch.Tensor:
        """Concatenate every rank's `t` along `dim`; rank r contributes sizes[r] entries along that axis (known to all
        ranks from the partition, so uneven bands are padded to the largest and trimmed)."""
        if self.world == 1:
            return t
        from mdtile import sharding
        assert t.shape[dim] == sizes[self.rank]
        big = max(sizes)
        src = t
        if t.shape[dim] < big:
            pad_shape = list(t.shape)
            pad_shape[dim] = big - t.shape[dim]
            src = torch.cat([t, t.new_zeros(pad_shape)], dim=dim)
        parts = sharding.comm_allgather(src, self.world, self.group)
        return torch.cat([p.narrow(dim, 0, sizes[r]) for r, p in enumerate(parts)], dim=dim)


class EngineOps:
    """The product's ops: thin calls into the mdtile C ABI (GPU only, no fallback)."""

    def __init__(self):
        import mdtile
        self.E = mdtile

    def ksize(self, conv) -> int:
        return conv.ksize

    def fuses_pre_gn(self, conv, upsample: bool) -> bool:
        return conv.fuses_pre_gn(upsample2x=upsample)

    def conv(self, conv, x, residual=None, upsample2x=False, pre_gn=None, token_major=False):
        return conv(x, residual=residual, upsample2x=upsample2x, token_major=token_major, pre_gn=pre_gn)

    def gn_sums(self, x, row_lo, row_hi):
        return self.E.gn_sums(x, row_lo, row_hi, 32)

    def gn_from_sums(self, sums, count):
        return self.E.gn_from_sums(sums, count)

    def gn_coeffs(self, mean, var, gamma, beta, C):
        return self.E.gn_coeffs(mean, var, gamma, beta, C, 32, 1e-6)

    def gn_apply(self, x, mean, var, gamma, beta, silu, inplace):
        return self.E.gn_apply(x, mean, var, gamma, beta, 32, 1e-6, silu, out=x if inplace else None)

    def attn_qk(self, q, k, v_tok, scale):
        return self.E.vae_attn_qk(q, k, v_tok, scale)

    def tanh(self, x):
        return self.E.tanh(x)


def estimate_group_norm_sp(steps: Sequence, zs: torch.Tensor, comm: BandComm, ops, fuse_pre_gn: bool = True
                           ) -> Optional[List[Tuple[torch.Tensor, torch.Tensor]]]:
    """Run the program `steps` (scripts/tilevae.py build_task_queue) on the down-sampled latent `zs` [B, 4, H, W] (the same
    tensor on every rank) split by rows across `comm.world` ranks, and return the frozen (var, mean) of every GroupNorm
    -- identical on all ranks, equal to the single-GPU estimator up to fp64 summation order.  None if a statistic is NaN
    (upstream falls back to slow mode, tilevae.py:500-503)."""
    B, _, H, W = zs.shape
    world, rank = comm.world, comm.rank
    if H < world:
        raise ValueError(f"sequence-parallel estimator: {H} rows cannot be split over {world} ranks")
    bounds = row_bounds(H, world)
    ht, hb = (1 if rank > 0 else 0), (1 if rank < world - 1 else 0)
    rows = bounds[rank + 1] - bounds[rank]
    x = zs[:, :, bounds[rank] - ht:bounds[rank + 1] + hb, :].contiguous()
    halo_ok = True            # halo slots hold the neighbours' rows of the CURRENT activation
    up = 1                    # resolution multiplier so far (rows of the full activation = H * up)
    res: list = []
    pre = None
    frozen: List[Tuple[torch.Tensor, torch.Tensor]] = []
    n_norm = sum(1 for s in steps if s.kind == "norm")

    for pc, s in enumerate(steps):
        if s.kind == "store_res":
            if s.conv is not None and ops.ksize(s.conv) == 3 and not halo_ok:    # 3x3 conv_shortcut reads the neighbours' rows too
                comm.exchange_halos(x, ht, hb, rows)
                halo_ok = True
            res.append(x if s.conv is None else ops.conv(s.conv, x))
        elif s.kind == "norm":
            C = x.shape[1]
            sums = comm.allreduce_sum(ops.gn_sums(x, ht, ht + rows))
            var, mean = ops.gn_from_sums(sums, float(C // 32) * float(H * up) * float(W))
            frozen.append((var, mean))
            if len(frozen) == n_norm:
                break
            if bool(torch.isnan(mean).any().item()) or bool(torch.isnan(var).any().item()):   # same value on every rank
                print("Nan detected in fast mode estimation. Fast mode disabled.")
                return None
            gamma, beta = s.norm
            nxt = steps[pc + 1] if pc + 1 < len(steps) else None
            if fuse_pre_gn and s.silu and nxt is not None and nxt.kind == "conv" and ops.fuses_pre_gn(nxt.conv, nxt.upsample):
                pre = ops.gn_coeffs(mean, var, gamma, beta, C)
            else:
                keep = bool(res) and res[-1] is x
                x = ops.gn_apply(x, mean, var, gamma, beta, s.silu, inplace=not keep)
        elif s.kind == "conv":
            k3 = ops.ksize(s.conv) == 3
            if k3 and not halo_ok:
                comm.exchange_halos(x, ht, hb, rows)
                halo_ok = True
            y = ops.conv(s.conv, x, residual=res.pop() if s.fuse_res else None, upsample2x=s.upsample, pre_gn=pre)
            pre = None
            if s.upsample:
                # each halo row became two rows: drop the outer (needs a row this rank does not have), keep the inner (exact)
                y = y[:, :, ht:y.shape[2] - hb, :].contiguous()
                rows, W, up = rows * 2, W * 2, up * 2
                halo_ok = True
            elif k3:
                halo_ok = False
            x = y
        elif s.kind == "attn":
            a = s.attn
            Bh, C, He, Wc = x.shape
            q = ops.conv(a.q, x)
            k = ops.conv(a.k, x)
            v = ops.conv(a.v, x, token_major=True)                      # [B, He*W, C]
            q_own = q[:, :, ht:ht + rows, :].reshape(Bh, C, rows * Wc).contiguous()
            k_own = k[:, :, ht:ht + rows, :].reshape(Bh, C, rows * Wc).contiguous()
            v_own = v[:, ht * Wc:(ht + rows) * Wc, :].contiguous()
            sizes = [(bounds[r + 1] - bounds[r]) * up * Wc for r in range(world)]
            k_all = comm.allgather_cat(k_own, 2, sizes)
            v_all = comm.allgather_cat(v_own, 1, sizes)
            o = ops.attn_qk(q_own, k_all.contiguous(), v_all.contiguous(), float(int(C) ** (-0.5)))
            oe = x.new_zeros((Bh, C, He, Wc))
            oe[:, :, ht:ht + rows, :] = o.view(Bh, C, rows, Wc)
            x = ops.conv(a.proj, oe, residual=res.pop())
            halo_ok = False
        elif s.kind == "tanh":
            x = ops.tanh(x)
    return frozen
