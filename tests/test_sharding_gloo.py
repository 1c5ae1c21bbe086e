"""Multi-GPU host logic on CPU: world_size 2 and 3 over the `gloo` backend (one process per rank, 127.0.0.1 rendezvous).

What is covered without a GPU: the band partition of the tile grid, the neighbour halo exchange of overlap-row partial sums
(mdtile/sharding.py:exchange_and_sum) and the slow-mode GroupNorm statistics all-reduce (allreduce_stats).  The per-rank
partial sums themselves are formed here by the ORACLE's arithmetic (on the GPU they come from mdtile_blend with
MDTILE_BLEND_PARTIAL | MDTILE_BLEND_TILE_RANGE, which tests/test_gpu_blend.py checks against the same oracle), so the
assertion is end-to-end: sharded blend == the single-process reference result on every rank's rows."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN = os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(rank, world, port):
    for p in (ROOT, PLUGIN):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)


def _partial_sums(o, x, band, N):
    """Raw fp32 sums of this band's tiles only (what mdtile_blend(PARTIAL | TILE_RANGE) writes)."""
    from oracle import blend_oracle as bo
    buf = torch.zeros_like(x)
    for t in range(band.tile_lo, band.tile_hi):
        bx, by, tw, th = o.boxes[t]
        out = bo.synthetic_denoiser(x[:, :, by:by + th, bx:bx + tw])
        if o.method == "md":
            buf[:, :, by:by + th, bx:bx + tw] += out
        else:
            buf[:, :, by:by + th, bx:bx + tw] += out * (o.tile_weights * o.rescale[:, :, by:by + th, bx:bx + tw])
    return buf


def _worker_blend(rank, world, port, method, W, H, tw, th, ov, q):
    try:
        _setup(rank, world, port)
        from mdtile import sharding
        from oracle import blend_oracle as bo
        o = bo.BlendOracle(method, W, H, tw, th, ov, 4)
        ys = sorted(set(b[1] for b in o.boxes))
        cols = len(set(b[0] for b in o.boxes))
        bands = sharding.band_partition(ys, o.th, cols, H, world)
        band = bands[rank]
        # every tile belongs to exactly one band; owned rows tile [0, H) exactly
        assert sum(b.tile_hi - b.tile_lo for b in bands) == len(o.boxes)
        live = [b for b in bands if not b.empty]
        assert live[0].own_lo == 0 and live[-1].own_hi == H and all(a.own_hi == b.own_lo for a, b in zip(live, live[1:]))
        torch.manual_seed(0)
        x = torch.randn(2, 4, H, W)
        # per-tile denoiser evaluation == the batched one for this synthetic map, so the single-process result is:
        ref = o.evaluate(x, bo.synthetic_denoiser)
        if not band.empty:
            part = _partial_sums(o, x, band, 2)
            sharding.exchange_and_sum(part, bands, rank)
            rows = slice(band.row_lo, band.row_hi)
            if method == "md":
                wgt = o.weights[:, :, rows]
                got = torch.where(wgt > 1, part[:, :, rows] / wgt, part[:, :, rows])
            else:
                got = part[:, :, rows]
            err = (got - ref[:, :, rows]).abs().max().item()
            # sums are re-associated across ranks (band partial + band partial), not bit-exact: fp32 round-off only
            assert err < 2e-6, f"rank {rank}: sharded blend differs from the single-process result by {err}"
            # both sides of a halo hold bit-identical sums (fixed ascending-rank accumulation order)
            for peer, lo, hi in sharding.halo_rows(bands, rank):
                mine = part[:, :, lo:hi].contiguous()
                theirs = torch.empty_like(mine)
                ops = [dist.P2POp(dist.isend, mine, peer), dist.P2POp(dist.irecv, theirs, peer)]
                for r in dist.batch_isend_irecv(ops):
                    r.wait()
                assert torch.equal(mine, theirs), f"rank {rank}<->{peer}: halo sums not bit-identical"
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _worker_stats(rank, world, port, q):
    try:
        _setup(rank, world, port)
        from mdtile import sharding
        g = torch.Generator().manual_seed(7)
        T, BG = 5, 32
        means, vars_ = torch.randn(T, BG, generator=g), torch.rand(T, BG, generator=g)
        px = torch.tensor([64.0 * 64, 64 * 86, 86 * 86, 70 * 64, 64 * 64])
        mine = list(range(rank, T, world))                       # == sharding.tiles_of_rank
        assert mine == sharding.tiles_of_rank(T, rank, world)
        sm = (means[mine] * px[mine, None]).sum(0)
        sv = (vars_[mine] * px[mine, None]).sum(0)
        var, mean = sharding.allreduce_stats(sm, sv, px[mine].sum().view(1))
        p = px / px.sum()                                         # GroupNormParam.summary pooling (tilevae.py:320-335)
        assert torch.allclose(mean, (means * p[:, None]).sum(0), rtol=1e-5, atol=1e-6)
        assert torch.allclose(var, (vars_ * p[:, None]).sum(0), rtol=1e-5, atol=1e-6)
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [f"rank {r}: {msg}" for r, msg in results if msg != "ok"]
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("world,method,W,H,tw,th,ov", [
    (2, "md", 160, 120, 48, 40, 16),
    (2, "mod", 160, 120, 48, 40, 16),
    (3, "md", 128, 256, 96, 96, 48),     # heavy overlap: halos reach past the direct neighbour
    (2, "mod", 256, 256, 128, 128, 8),   # cfg4-shaped grid (scaled down)
    (3, "md", 64, 64, 48, 48, 8),        # fewer tile rows (2) than ranks: one rank idles
])
def test_sharded_blend_halo_exchange(world, method, W, H, tw, th, ov):
    _run(_worker_blend, world, method, W, H, tw, th, ov)


def test_slow_mode_stats_allreduce():
    _run(_worker_stats, 2)


def _worker_bringup(rank, world, port, q):
    """No GPU here: the C-ABI context cannot come up, so the checked bring-up must report 'inactive' on EVERY rank (the vote)
    and leave exchange_and_sum on the torch.distributed path, still exact."""
    try:
        _setup(rank, world, port)
        from mdtile import sharding
        active = sharding.init_process_context_checked(rank, world, 0, timeout_s=30.0)
        assert active is False and sharding._CTX is None
        rows = 8 * world
        bands = [sharding.Band(r, r, r + 1, r, r + 1, max(0, 8 * r - 3), min(rows, 8 * r + 11), 8 * r, 8 * r + 8) for r in range(world)]
        parts = [torch.randn(2, 4, rows, 16, generator=torch.Generator().manual_seed(90 + r)) for r in range(world)]
        mine = parts[rank].clone()
        sharding.exchange_and_sum(mine, bands, rank)
        lo, hi = bands[rank].row_lo, bands[rank].row_hi
        want = torch.zeros_like(mine)
        for r in range(world):
            a, e = max(lo, bands[r].row_lo), min(hi, bands[r].row_hi)
            if a < e:
                want[:, :, a:e] += parts[r][:, :, a:e]
        assert torch.equal(mine[:, :, lo:hi], want[:, :, lo:hi])
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_checked_bringup_votes_and_falls_back(world):
    _run(_worker_bringup, world)


def test_cost_aware_tile_deal():
    """mdtile/sharding.py: deal_tiles -- VAE tiles by area (longest-processing-time greedy), the same owner list on every rank.  The 8K
    decode's 4 x 4 grid (nine 278 x 278, three 278 x 256, three 256 x 278, one 256 x 256 latent tiles) on 4 ranks: round-robin gives ranks
    0-2 a whole column of large tiles each (2 % over the mean on the heaviest rank), the deal 3-2-2-2 (0.1 %); on 8 ranks every rank gets
    two tiles (nine large ones: one pair of them is unavoidable); equal tiles degenerate to round-robin."""
    from mdtile import sharding
    from oracle import vae_oracle as vo
    ins, _ = vo.split_tiles(1024, 1024, 256)
    area = [(b[1] - b[0]) * (b[3] - b[2]) for b in ins]
    assert sorted(area, reverse=True)[:9] == [278 * 278] * 9 and len(ins) == 16
    mean4 = sum(area) / 4
    rr = max(sum(area[i] for i in range(r, 16, 4)) for r in range(4))
    own = sharding.deal_tiles(ins, 4)
    loads = [sum(a for a, o in zip(area, own) if o == r) for r in range(4)]
    assert sorted(sum(1 for a, o in zip(area, own) if o == r and a == 278 * 278) for r in range(4)) == [2, 2, 2, 3]
    assert max(loads) / mean4 < 1.002 < 1.015 < rr / mean4
    own8 = sharding.deal_tiles(ins, 8)
    assert sorted(own8) == sorted(list(range(8)) * 2)
    assert max(sum(a for a, o in zip(area, own8) if o == r) for r in range(8)) == 2 * 278 * 278
    same = [[0, 64, 0, 64]] * 7
    assert sharding.deal_tiles(same, 3) == [0, 1, 2, 0, 1, 2, 0]
    assert sharding.deal_tiles(ins, 1) == [0] * 16
