"""
Multi-GPU sharding of the hot path: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  Nothing upstream corresponds to this file -- the reference is single-process/single-device.

Blend (one model evaluation):
    tiles are split into contiguous ROW BANDS of the tile grid, one band per rank.  A rank gathers / evaluates only its own
    tiles and forms fp32 partial sums for the canvas rows those tiles touch (mdtile_blend with MDTILE_BLEND_PARTIAL |
    MDTILE_BLEND_TILE_RANGE | a row range).  Only the rows where two neighbouring bands overlap need data from another
    rank: each rank swaps exactly those `overlap_rows x W x N*C` fp32 slabs with its upper/lower neighbour (a pairwise
    all-reduce expressed as grouped send/recv: xGMI is point-to-point, so two neighbour transfers beat a ring all-reduce
    of the 33.5 MB canvas by ~30x), adds them in a FIXED order (lower rank's partial first, so both sides get bit-identical
    sums), and finalises its rows (mdtile_blend_finalize).  Every rank ends up with the finished rows its own tiles need for
    the next evaluation -- no second exchange, no full-canvas collective.

VAE decode: tiles are dealt round-robin; the fast-mode statistics estimator (one untiled pass) is split by rows across the
    ranks (mdtile/seqpar.py; MDTILE_SP_ESTIMATOR=0: every rank repeats it, bit-identical, no communication); slow mode
    all-reduces 2*B*32+1 floats per GroupNorm barrier.  Output tiles are disjoint; they stay sharded unless the caller
    gathers them.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclass
class Band:
    rank: int
    tile_row_lo: int      # first tile row of the band
    tile_row_hi: int      # one past the last tile row
    tile_lo: int          # first tile index (row-major)
    tile_hi: int
    row_lo: int           # canvas rows touched by the band's tiles: [row_lo, row_hi)
    row_hi: int
    own_lo: int           # canvas rows this rank is the OWNER of (disjoint cover of [0, H)), used for gathers
    own_hi: int

    @property
    def empty(self) -> bool:
        return self.tile_row_hi <= self.tile_row_lo


def band_partition(ys: Sequence[int], tile_h: int, cols: int, H: int, world: int) -> List[Band]:
    """Split `len(ys)` tile rows into `world` contiguous bands (sizes differ by at most one; trailing ranks may be empty
    when there are fewer tile rows than ranks)."""
    rows = len(ys)
    base, extra = divmod(rows, world)
    bands, r0 = [], 0
    for rank in range(world):
        n = base + (1 if rank < extra else 0)
        r1 = r0 + n
        if n > 0:
            row_lo, row_hi = ys[r0], ys[r1 - 1] + tile_h
        else:
            row_lo = row_hi = H
        bands.append(Band(rank, r0, r1, r0 * cols, r1 * cols, row_lo, row_hi, 0, 0))
        r0 = r1
    # ownership: a canvas row belongs to the LAST band that touches it... any disjoint rule works; use band starts
    live = [b for b in bands if not b.empty]
    for i, b in enumerate(live):
        b.own_lo = 0 if i == 0 else b.row_lo
        b.own_hi = H if i == len(live) - 1 else live[i + 1].row_lo
    return bands


def halo_rows(bands: Sequence[Band], rank: int) -> List[Tuple[int, int, int]]:
    """[(peer_rank, row_lo, row_hi)]: canvas row ranges this rank's partial sums share with another band.  With
    overlap < tile_h/2 only direct neighbours appear; heavier overlaps (e.g. 48 of 96 with one tile row per rank) can reach
    further, which this handles by intersecting with every other band."""
    me = bands[rank]
    out = []
    if me.empty:
        return out
    for other in bands:
        if other.rank == rank or other.empty:
            continue
        lo, hi = max(me.row_lo, other.row_lo), min(me.row_hi, other.row_hi)
        if lo < hi:
            out.append((other.rank, lo, hi))
    return out


def exchange_and_sum(partial: torch.Tensor, bands: Sequence[Band], rank: int, group=None) -> torch.Tensor:
    """Swap overlap-row slabs of `partial` ([N, C, H, W] fp32, valid on this band's rows) with the peers that share them and
    accumulate in rank order, in place.  Returns `partial` (complete on [row_lo, row_hi))."""
    halos = halo_rows(bands, rank)
    if not halos:
        return partial
    # RCCL ("nccl") moves device memory directly; a backend that cannot (gloo: CPU tests, single-GPU multi-process checks)
    # gets the slabs staged through the host
    staged = partial.is_cuda and dist.get_backend(group) == "gloo"
    send = {peer: partial[:, :, lo:hi, :].contiguous() for peer, lo, hi in halos}
    if staged:
        send = {peer: t.cpu() for peer, t in send.items()}
    recv = {peer: torch.empty_like(send[peer]) for peer, _, _ in halos}
    ops = []
    for peer, _, _ in halos:
        ops.append(dist.P2POp(dist.isend, send[peer], peer, group=group))
        ops.append(dist.P2POp(dist.irecv, recv[peer], peer, group=group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if staged:
        send = {peer: t.to(partial.device) for peer, t in send.items()}
        recv = {peer: t.to(partial.device) for peer, t in recv.items()}
    # deterministic, rank-symmetric summation order: contributions are added in ascending rank order on BOTH sides.
    # rows can be shared by more than two bands, so build each shared row range from all contributors.
    events = sorted(set([r for _, lo, hi in halos for r in (lo, hi)]))
    for a, b in zip(events[:-1], events[1:]):
        contributors = sorted([rank] + [peer for peer, lo, hi in halos if lo <= a and b <= hi])
        if len(contributors) == 1:
            continue
        acc = None
        for c in contributors:
            if c == rank:
                piece = send_piece(partial, a, b)
            else:
                plo = [lo for peer, lo, hi in halos if peer == c][0]
                piece = recv[c][:, :, a - plo:b - plo, :]
            acc = piece.clone() if acc is None else acc.add_(piece)
        partial[:, :, a:b, :] = acc
    return partial


def send_piece(partial: torch.Tensor, a: int, b: int) -> torch.Tensor:
    return partial[:, :, a:b, :]


def tiles_of_rank(num_tiles: int, rank: int, world: int) -> List[int]:
    """VAE tiles dealt round-robin (tiles differ in size only at the image border, so this balances well)."""
    return list(range(rank, num_tiles, world))


def allreduce_stats(sum_mean_px: torch.Tensor, sum_var_px: torch.Tensor, px: torch.Tensor, group=None):
    """Slow-mode GroupNorm barrier across ranks: all-reduce(sum) of [sum_i px_i*mean_i, sum_i px_i*var_i, sum_i px_i]."""
    buf = torch.cat([sum_mean_px.flatten(), sum_var_px.flatten(), px.flatten()])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    n = sum_mean_px.numel()
    total = buf[2 * n]
    return (buf[n:2 * n] / total).view_as(sum_var_px), (buf[:n] / total).view_as(sum_mean_px)
