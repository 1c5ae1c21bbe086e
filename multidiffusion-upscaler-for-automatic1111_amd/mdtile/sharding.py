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
import os
from typing import List, Optional, Sequence, Tuple

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
    if _use_ctx(partial, group):
        return exchange_and_sum_ctx(partial, bands)      # C ABI: pack -> ncclSend / ncclRecv -> k_halo_add
    # torch.distributed path (CPU tests over gloo, single-GPU multi-process checks).  RCCL ("nccl") moves device memory directly; a backend that cannot (gloo: CPU tests, single-GPU multi-process checks)
    # gets the slabs staged through the host
    staged = _staged(partial, group)
    group = _grp(group)
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


def deal_tiles(in_bboxes: Sequence[Sequence[int]], world: int) -> List[int]:
    """owner[i] of every VAE tile, by COST: the tiles of one image come in up to four sizes (interior tiles carry padding on both
    sides of an axis, border tiles on one), and round-robin hands whole tile COLUMNS of one size to a rank whenever the grid width is
    a multiple of the rank count -- the 8K decode's 4 x 4 grid on 4 ranks puts three 278 x 278 tiles + one 278 x 256 on ranks 0-2 and
    the three 256-wide ones + the 256 x 256 corner on rank 3 (2 % over the mean on the heaviest rank).  Longest-processing-time
    greedy on the tile areas (decode cost is linear in input pixels): tiles by decreasing area (ties: index order), each to the rank
    with the least pixels so far (ties: lowest rank) -- 3-2-2-2 of the nine large tiles there, 0.1 % over the mean.  Deterministic,
    computed identically on every rank from the bboxes alone; equal-size tiles degenerate to round-robin."""
    order = sorted(range(len(in_bboxes)), key=lambda i: (-(in_bboxes[i][1] - in_bboxes[i][0]) * (in_bboxes[i][3] - in_bboxes[i][2]), i))
    load = [0] * world
    owner = [0] * len(in_bboxes)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += (in_bboxes[i][1] - in_bboxes[i][0]) * (in_bboxes[i][3] - in_bboxes[i][2])
    return owner


def allreduce_stats(sum_mean_px: torch.Tensor, sum_var_px: torch.Tensor, px: torch.Tensor, group=None):
    """Slow-mode GroupNorm barrier across ranks: all-reduce(sum) of [sum_i px_i*mean_i, sum_i px_i*var_i, sum_i px_i]."""
    buf = torch.cat([sum_mean_px.flatten(), sum_var_px.flatten(), px.flatten()])
    buf = comm_allreduce_sum(buf, group)
    n = sum_mean_px.numel()
    total = buf[2 * n]
    return (buf[n:2 * n] / total).view_as(sum_var_px), (buf[:n] / total).view_as(sum_mean_px)


# ---------------------------------------------------------------------------------------------------------------------
# C-ABI shard context (include/mdtile.h "Multi-GPU"): the halo exchange as pack -> grouped ncclSend / ncclRecv -> fixed-order
# k_halo_add, three launches per evaluation instead of the eager slab arithmetic of exchange_and_sum above; and the generic
# collectives of the VAE side (row halos, statistics all-reduce, K / V all-gather, image gather) on the same communicator.
# ---------------------------------------------------------------------------------------------------------------------
_CTX = None          # mdtile.Shard of THIS process when it is one rank of a process-per-GPU job
_CTX_SCRATCH = {}
_DATA_GROUP = None   # torch.distributed group of the data plane when there is no C-ABI context (None = the default group)


def process_context():
    return _CTX


def set_data_group(group) -> None:
    """The torch.distributed group that carries tensors when the job has no C-ABI context: e.g. ONE "nccl" group next to a gloo
    default group that only does control traffic (bench.py)."""
    global _DATA_GROUP
    _DATA_GROUP = group


def _grp(group):
    return group if group is not None else _DATA_GROUP


def _use_ctx(t: torch.Tensor, group) -> bool:
    return _CTX is not None and group is None and t.is_cuda


def _staged(t: torch.Tensor, group) -> bool:
    """torch.distributed path with a backend that cannot move device memory (gloo): go through the host."""
    return t.is_cuda and dist.get_backend(_grp(group)) == "gloo"


def comm_p2p(ops, group=None) -> None:
    """Grouped point-to-point of this rank: ops = [(peer, tensor to send or None, tensor to receive into or None)], contiguous.
    C-ABI context (one ncclGroup on torch's current stream) when active, torch.distributed batch_isend_irecv otherwise."""
    ops = [o for o in ops if o[1] is not None or o[2] is not None]
    if not ops:
        return
    probe = next(t for o in ops for t in o[1:] if t is not None)
    if _use_ctx(probe, group):
        _CTX.p2p([ops], streams=[torch.cuda.current_stream(probe.device).cuda_stream])
        return
    staged = _staged(probe, group)
    group = _grp(group)
    reqs, back = [], []
    for peer, snd, rcv in ops:
        if snd is not None:
            reqs.append(dist.P2POp(dist.isend, snd.cpu() if staged else snd, peer, group=group))
        if rcv is not None:
            h = torch.empty(rcv.shape, dtype=rcv.dtype) if staged else rcv
            reqs.append(dist.P2POp(dist.irecv, h, peer, group=group))
            if staged:
                back.append((rcv, h))
    for r in dist.batch_isend_irecv(reqs):
        r.wait()
    for rcv, h in back:
        rcv.copy_(h)


def comm_allreduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """All-reduce(sum) in place (returns t).  The C-ABI context reduces fp64; other dtypes are widened for the trip."""
    if _use_ctx(t, group):
        d = t if (t.dtype == torch.float64 and t.is_contiguous()) else t.double().contiguous()
        _CTX.allreduce_stats([d], streams=[torch.cuda.current_stream(t.device).cuda_stream])
        if d is not t:
            t.copy_(d)
        return t
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=_grp(group))
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_grp(group))
    return t


def comm_allgather(t: torch.Tensor, world: int, group=None) -> List[torch.Tensor]:
    """Every rank's `t` (one common shape), in rank order."""
    t = t.contiguous()
    if _use_ctx(t, group):
        return list(_CTX.allgather([t], streams=[torch.cuda.current_stream(t.device).cuda_stream])[0].unbind(0))
    src = t.cpu() if _staged(t, group) else t
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=_grp(group))
    return [p.to(t.device) if p.device != t.device else p for p in parts]


def gather_tiles_to_root(result: torch.Tensor, out_bboxes: Sequence[Sequence[int]], owner_of, rank: int, root: int = 0, group=None) -> torch.Tensor:
    """VAE output tiles are disjoint rectangles of the result canvas, each decoded by one rank (`owner_of(i)`).  Every foreign
    rectangle travels to `root` in ONE grouped exchange (each pair of GPUs has its own xGMI link, so the sends of different
    owners run concurrently) and is pasted into root's canvas -- the single tensor upstream's vae_tile_forward returns
    (scripts/tilevae.py:630-656)."""
    ops, paste = [], []
    for i, (x1, x2, y1, y2) in enumerate(out_bboxes):
        own = owner_of(i)
        if own == root:
            continue
        if rank == own:
            ops.append((root, result[:, :, y1:y2, x1:x2].contiguous(), None))
        elif rank == root:
            buf = torch.empty((result.shape[0], result.shape[1], y2 - y1, x2 - x1), dtype=result.dtype, device=result.device)
            ops.append((own, None, buf))
            paste.append(((x1, x2, y1, y2), buf))
    comm_p2p(ops, group)
    for (x1, x2, y1, y2), buf in paste:
        result[:, :, y1:y2, x1:x2].copy_(buf)
    return result


def init_process_context(rank: int, world: int, device: int, group=None):
    """One rank per process (bench.py under torchrun): rank 0 draws the RCCL unique id, torch.distributed carries it to the
    others, every rank joins the communicator through the C ABI.  Afterwards exchange_and_sum / allreduce_stats run on it."""
    global _CTX
    import mdtile
    box = [mdtile.Shard.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    _CTX = mdtile.Shard(nranks=world, rank=rank, uid=box[0], device=device)
    return _CTX


def _run_with_timeout(fn, timeout_s: float, device=None):
    """fn() in a worker thread.  Returns (value, error text or None).  A call that does not return in time is abandoned: a
    result that arrives late is destroyed by the worker itself and never reaches the caller."""
    import threading
    box = {"abandoned": False}

    def _work():
        try:
            if device is not None and torch.cuda.is_available():
                torch.cuda.set_device(device)       # the current device is per thread
            v = fn()
            if box["abandoned"] and hasattr(v, "destroy"):
                v.destroy()
            else:
                box["value"] = v
        except BaseException as e:      # noqa: BLE001 -- reported to the caller, the job continues on the fallback path
            box["error"] = e

    t = threading.Thread(target=_work, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        box["abandoned"] = True
        return None, "timed out"
    if "error" in box:
        return None, repr(box["error"])
    return box.get("value"), None


def _vote(ok: bool, dev, group) -> bool:
    """All ranks agree (control group: a host tensor on gloo, a device tensor on an RCCL-backed group)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if dist.get_backend(group) == "gloo" else dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


def init_process_context_checked(rank: int, world: int, device: int, timeout_s: float = 120.0, group=None) -> bool:
    """init_process_context with a seat belt for its first contact with a given node.  Every collective of the CONTROL group
    (`group`, torch.distributed) is issued from the main thread, in the same order on every rank:
      1. rank 0 draws the RCCL id (a failure travels as a sentinel) and broadcasts it;
      2. ONLY ncclCommInitRank runs in a worker thread (a rendezvous that never completes must not hang the job); vote;
      3. the engine's own bring-up check (mdtile_shard_selfcheck: all-reduce, broadcast, ring send / receive, all-gather) and a halo
         exchange compared with the same exchange over torch.distributed, again under a timeout; vote;
    The context becomes visible (module global, main thread) only after every rank voted yes twice; a communicator that arrives
    late is destroyed, never used.  Returns whether the C-ABI path is active; otherwise the job stays on torch.distributed."""
    global _CTX
    import sys
    import mdtile
    _CTX = None          # the reference exchange of step 3 must take the torch.distributed path, whatever an earlier call left installed
    check_scratch: dict = {}      # private to this bring-up (see exchange_and_sum_ctx)
    dev = torch.device("cuda", device) if torch.cuda.is_available() else torch.device("cpu")   # (cpu: the gloo tests of this logic)
    box = [None]
    if rank == 0:
        try:
            box[0] = (mdtile.Shard.unique_id(), mdtile.Shard.unique_id())      # (probe communicator, real communicator)
        except BaseException as e:      # noqa: BLE001
            box[0] = ("error", repr(e))
    dist.broadcast_object_list(box, src=0, group=group)
    ctx, why = None, None
    ids = box[0] if isinstance(box[0], tuple) and len(box[0]) == 2 and isinstance(box[0][0], (bytes, bytearray)) else None
    if ids is not None:
        # 2a. the rendezvous itself, interruptibly: a NON-BLOCKING probe communicator polled under the deadline and aborted on it
        #     (mdtile_shard_probe_rank) -- a rendezvous that cannot complete on this node is found out with no thread left inside RCCL.
        #     (librccl without the non-blocking calls: the probe reports that and the blocking path below keeps its worker-thread seat belt.)
        probe_ok = True
        if dev.type == "cuda" and os.environ.get("MDTILE_SHARD_PROBE", "1") != "0":
            _, perr = _run_with_timeout(lambda: mdtile.Shard.probe(world, rank, bytes(ids[0]), device, timeout_s=min(timeout_s, 60.0)) or True,
                                        timeout_s + 10.0, device)
            if perr is not None and "has no ncclCommInitRankConfig" not in perr:
                probe_ok, why = False, f"bring-up probe: {perr}"
        probe_ok = _vote(probe_ok, dev, group)
        # 2b. the real (blocking) communicator, only when the probe came up on EVERY rank
        if probe_ok:
            ctx, why = _run_with_timeout(lambda: mdtile.Shard(nranks=world, rank=rank, uid=bytes(ids[1]), device=device), timeout_s, device)
        elif why is None:
            why = "bring-up probe failed on another rank"
    else:
        why = f"rank 0 could not draw an RCCL id: {box[0][1] if box[0] else 'no id'}"
    ok = _vote(ctx is not None, dev, group)
    if ok:
        rows = 8 * world
        bands = [Band(r, r, r + 1, r, r + 1, max(0, 8 * r - 2), min(rows, 8 * r + 10), 8 * r, 8 * r + 8) for r in range(world)]

        def _payload():
            return torch.randn(2, 4, rows, 64, generator=torch.Generator(device="cpu").manual_seed(1234 + rank)).to(dev)

        def _check():
            ctx.selfcheck()
            a = _payload()
            exchange_and_sum_ctx(a, bands, ctx, scratch=check_scratch)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            return a
        a, why = _run_with_timeout(_check, timeout_s, device)
        # the reference exchange is a collective of the control group: EVERY rank runs it, whatever its own check said
        b = exchange_and_sum(_payload(), bands, rank, group)        # torch.distributed path: the context is not installed yet
        good = a is not None
        if good:
            lo, hi = bands[rank].row_lo, bands[rank].row_hi
            if not torch.equal(a[:, :, lo:hi], b[:, :, lo:hi]):
                good, why = False, "halo self-check mismatch against torch.distributed"
        ok = _vote(good, dev, group)
    if why:
        print(f"[mdtile] rank {rank}: C-ABI shard context unavailable ({why}); collectives stay on torch.distributed", file=sys.stderr)
    _CTX_SCRATCH.clear()
    if ok:
        _CTX = ctx
        return True
    _CTX = None
    if ctx is not None:
        try:
            _run_with_timeout(ctx.destroy, 10.0, device)
        except BaseException:   # noqa: BLE001
            pass
    return False


def band_rows_table(bands: Sequence[Band]) -> List[int]:
    out = []
    for b in bands:
        out += [b.row_lo, b.row_hi] if not b.empty else [0, 0]
    return out


def exchange_and_sum_ctx(partial: torch.Tensor, bands: Sequence[Band], ctx=None, scratch: Optional[dict] = None) -> torch.Tensor:
    """exchange_and_sum on a shard context holding ONE local rank (this process): in place, on torch's current stream.
    scratch: the cache of halo scratch buffers to use (default: the module's, which belongs to the installed context -- the bring-up
    check passes its own, so that a worker thread abandoned after a timeout can never touch the one the job goes on to use)."""
    ctx = ctx or _CTX
    cache = _CTX_SCRATCH if scratch is None else scratch
    table = band_rows_table(bands)
    key = (tuple(table), tuple(partial.shape), partial.device.index)
    if key not in cache:
        cache.clear()
        cache[key] = ctx.halo_scratch(table, partial.shape[0], partial.shape[1], partial.shape[3])
    ctx.halo_exchange([partial], cache[key], table, streams=[torch.cuda.current_stream(partial.device).cuda_stream])
    return partial


class ShardedBlend:
    """One model evaluation (tile gather + overlap blend) split in row bands over the ranks of a mdtile.Shard -- all of them in
    this process (`mdtile.Shard(dev_ids=[...])`, the form a webui can use) or one per process.  Regions (BASELINE cfg5: region
    prompt control on 2 GPUs) are dealt round-robin to the ranks: the owner evaluates the region, its output is broadcast to the
    bands; a background region is ACCUMULATED only by the rank that owns the canvas row (its rows are cut out of the region output,
    so the shared rows are not counted twice), a foreground region is composited by every rank on the rows it finalises.

        sb = ShardedBlend(shard, W, H, tile_w, tile_h, overlap, tile_bs, method, regions=[(x, y, w, h, mode, feather_ratio)])
        outs = sb.step(xs, tile_fn, region_fn)      xs / outs: one [N, C, H, W] canvas per LOCAL rank, on that rank's device
    After step() canvas i is complete on the rows band i touches (what its tiles need for the next evaluation); allgather_rows()
    completes all of them (single-process contexts)."""

    def __init__(self, shard, W: int, H: int, tile_w: int, tile_h: int, overlap: int, tile_bs: int, method: int, regions=()):
        import mdtile
        self.E, self.shard, self.method, self.W, self.H = mdtile, shard, method, W, H
        self.regions = list(regions)
        self.local = []
        for i, dev in enumerate(shard.devices):
            with torch.cuda.device(dev):
                d = torch.device("cuda", dev)
                plan = mdtile.Plan(W, H, tile_w, tile_h, overlap, tile_bs)
                weights = torch.zeros(1, 1, H, W, device=d)
                tile_wt = mdtile.gaussian_weights(plan.tile_w, plan.tile_h, d) if method == mdtile.METHOD_MOD else None
                mdtile.weight_map_add_grid(plan, tile_wt, weights)
                rw = []       # per region: MoD background weight map / feather mask on this device
                for (x, y, w, h, mode, fr) in self.regions:
                    if mode == mdtile.REGION_BG:
                        cw = mdtile.gaussian_weights(w, h, d) if method == mdtile.METHOD_MOD else None
                        mdtile.weight_map_add_rect(weights, x, y, w, h, cw, 1.0)
                        rw.append(cw)
                    else:
                        rw.append(mdtile.feather_mask(w, h, fr, d))
                rescale = None
                if method == mdtile.METHOD_MOD:
                    rescale = mdtile.reciprocal(weights)
                    for k, (x, y, w, h, mode, fr) in enumerate(self.regions):
                        if mode == mdtile.REGION_BG:
                            mdtile.rect_mul_canvas(rw[k], rescale, x, y, w, h)
                self.local.append(dict(dev=d, plan=plan, weights=weights, tile_wt=tile_wt, rescale=rescale, rw=rw, partial=None, out=None, packed=None))
        plan0 = self.local[0]["plan"]
        ys = sorted(set(b[1] for b in plan0.bboxes))
        self.bands = band_partition(ys, plan0.tile_h, plan0.cols, H, shard.nranks)
        self.table = band_rows_table(self.bands)
        self.scratch = None

    def _stream(self, i):
        return torch.cuda.current_stream(self.local[i]["dev"]).cuda_stream

    def region_owner(self, k: int) -> int:
        return k % self.shard.nranks

    def step(self, xs, tile_fn, region_fn=None):
        E, sh = self.E, self.shard
        N, C = xs[0].shape[:2]
        if self.scratch is None:
            self.scratch = sh.halo_scratch(self.table, N, C, self.W)
        # 1. regions: evaluated by their owner, broadcast to every rank
        routs = [[None] * len(self.regions) for _ in self.local]
        for k, (x, y, w, h, mode, fr) in enumerate(self.regions):
            bufs = []
            for i, L in enumerate(self.local):
                with torch.cuda.device(L["dev"]):
                    if sh.first + i == self.region_owner(k):
                        t = region_fn(E.gather_rect(xs[i].contiguous(), x, y, w, h), k).to(xs[i].dtype).contiguous()
                    else:
                        t = torch.empty((N, C, h, w), dtype=xs[i].dtype, device=L["dev"])
                    bufs.append(t)
                    routs[i][k] = t
            sh.bcast(bufs, self.region_owner(k), streams=[self._stream(i) for i in range(len(self.local))])
        # 2. per rank: gather its tiles, evaluate, accumulate partial sums on the rows they touch
        for i, L in enumerate(self.local):
            band = self.bands[sh.first + i]
            with torch.cuda.device(L["dev"]):
                x = xs[i].contiguous()
                plan = L["plan"]
                if L["partial"] is None:
                    L["partial"] = torch.zeros(N, C, self.H, self.W, device=L["dev"])
                    L["out"] = torch.empty(N, C, self.H, self.W, dtype=x.dtype, device=L["dev"])
                    L["packed"] = torch.zeros(plan.num_tiles * N, C, plan.tile_h, plan.tile_w, dtype=x.dtype, device=L["dev"])
                if band.empty:
                    continue
                E.gather_range(plan, x, L["packed"], band.tile_lo, band.tile_hi)
                rows = slice(band.tile_lo * N, band.tile_hi * N)
                L["packed"][rows] = tile_fn(L["packed"][rows]).to(x.dtype)
                specs = []
                for k, (rx, ry, rw_, rh, mode, fr) in enumerate(self.regions):
                    if mode != E.REGION_BG:
                        continue
                    lo, hi = max(ry, band.own_lo), min(ry + rh, band.own_hi)       # accumulate a background region once: on OWNED rows
                    if lo >= hi:
                        continue
                    sub = E.gather_rect(routs[i][k], 0, lo - ry, rw_, hi - lo)
                    wsub = None
                    if self.method == E.METHOD_MOD:
                        wsub = L["rw"][k][lo - ry:hi - ry].contiguous()
                    specs.append(E.RegionSpec(rx, lo, rw_, hi - lo, E.REGION_BG, sub, wsub))
                kw = dict(weights=L["weights"]) if self.method == E.METHOD_MD else dict(tile_w=L["tile_wt"], rescale=L["rescale"])
                E.blend(plan, self.method, [L["packed"]], N, C, out=L["partial"], packed=True, partial=True, regions=specs,
                        tile_range=(band.tile_lo, band.tile_hi), row_range=(band.row_lo, band.row_hi), **kw)
        # 3. rows shared by two bands: swap + sum in rank order (C ABI: pack, ncclSend / ncclRecv, k_halo_add)
        sh.halo_exchange([L["partial"] for L in self.local], self.scratch, self.table, streams=[self._stream(i) for i in range(len(self.local))])
        # 4. epilogue on the band's rows: MD division, foreground composite
        outs = []
        for i, L in enumerate(self.local):
            band = self.bands[sh.first + i]
            with torch.cuda.device(L["dev"]):
                if not band.empty:
                    fg = [E.RegionSpec(rx, ry, rw_, rh, E.REGION_FG, routs[i][k], L["rw"][k])
                          for k, (rx, ry, rw_, rh, mode, fr) in enumerate(self.regions) if mode == E.REGION_FG]
                    E.blend_finalize(L["plan"], self.method, L["partial"], weights=L["weights"] if self.method == E.METHOD_MD else None,
                                     regions=fg, out=L["out"], dtype=L["out"].dtype, row_range=(band.row_lo, band.row_hi))
                outs.append(L["out"])
        return outs

    def allgather_rows(self, outs):
        """Single-process contexts: copy every band's OWNED rows to the other local canvases (peer copies over xGMI)."""
        assert self.shard.nlocal == self.shard.nranks, "allgather_rows needs every rank in this process"
        for i, L in enumerate(self.local):
            b = self.bands[i]
            if b.empty:
                continue
            for j in range(len(self.local)):
                if j != i:
                    outs[j][:, :, b.own_lo:b.own_hi].copy_(outs[i][:, :, b.own_lo:b.own_hi], non_blocking=True)
        for L in self.local:
            torch.cuda.current_stream(L["dev"]).synchronize()
        return outs
