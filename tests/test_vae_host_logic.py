"""Host logic of the Tiled VAE hook on CPU: the product's VAEHook (scripts/tilevae.py) driven with torch doubles of the
engine (tests/torch_engine.py) must reproduce the oracle -- tile scheduling, fast / slow / semi-fast (color_fix) GroupNorm
handling, crop + assemble, both directions; and with world_size 2 over gloo: tiles dealt across ranks, the sequence-parallel
estimator in fast mode and the pooled-statistics all-reduce in slow mode (each rank's own output tiles vs the oracle)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN = os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")
for _p in (ROOT, PLUGIN, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def _hook(net, ts, is_decoder, fast, color_fix=False):
    from hostsim import stub_host as sh
    import torch_engine as te
    sh.install("cpu")
    pl = sh.load_plugin()
    net.original_forward = net.forward
    hook = pl.tilevae.VAEHook(net, ts, is_decoder=is_decoder, fast_decoder=fast, fast_encoder=fast, color_fix=color_fix)
    hook.engine, hook._pack, hook._sp_ops = te.TorchEngine(), te.TorchConv, te.TorchSeqParOps()   # torch doubles of the engine
    return hook


@pytest.mark.parametrize("fast", [True, False])
def test_decode_matches_oracle(fast):
    from hostsim import ldm_decoder as ld
    from oracle import vae_oracle as vo
    dec = ld.make_decoder(0, small=True)
    torch.manual_seed(2)
    z = torch.randn(1, 4, 36, 44)
    with torch.no_grad():
        out = _hook(dec, 16, True, fast)(z)
        ref = vo.tiled_forward(ld.make_decoder(0, small=True), z, 16, fast)
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("fast,color_fix", [(True, False), (False, False), (True, True)])
def test_encode_matches_oracle(fast, color_fix):
    from hostsim import ldm_decoder as ld
    from oracle import vae_oracle as vo
    enc = ld.make_encoder(0, small=True)
    torch.manual_seed(4)
    x = torch.randn(1, 3, 136, 200)
    with torch.no_grad():
        out = _hook(enc, 64, False, fast, color_fix)(x)
        ref = vo.tiled_forward(ld.make_encoder(0, small=True), x, 64, fast, is_decoder=False, color_fix=color_fix)
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fast, q, hw=(40, 56)):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(1)
        from hostsim import ldm_decoder as ld
        from oracle import vae_oracle as vo
        dec = ld.make_decoder(0, small=True)
        torch.manual_seed(2)
        z = torch.randn(1, 4, *hw)
        hook = _hook(dec, 16, True, fast)
        hook.shard = (rank, world)
        hook.gather_to = 0          # rank 0 returns the assembled image (one grouped exchange of the tile rectangles)
        with torch.no_grad():
            out = hook(z)
            ref = vo.tiled_forward(ld.make_decoder(0, small=True), z, 16, fast)
        if rank == 0:
            err = (out - ref).abs().max().item() / ref.abs().max().item()
            assert err < 2e-4, f"assembled image on rank 0: rel err {err}"
        ins, outs = vo.split_tiles(hw[0], hw[1], 16, True)
        from mdtile import sharding
        owner = sharding.deal_tiles(ins, world)
        mine = [i for i in range(len(ins)) if owner[i] == rank]
        assert mine or hw != (40, 56), "the default geometry must give every rank a tile"
        for i in mine:
            ob = outs[i]
            a, b = out[:, :, ob[2]:ob[3], ob[0]:ob[1]], ref[:, :, ob[2]:ob[3], ob[0]:ob[1]]
            err = (a - b).abs().max().item() / ref.abs().max().item()
            assert err < 2e-4, f"rank {rank} tile {i}: rel err {err}"
        dist.barrier()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world,fast", [(2, True), (2, False), (3, True)])
def test_sharded_decode_over_gloo(world, fast):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fast, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [f"rank {r}: {msg}" for r, msg in results if msg != "ok"]
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("fast", [True, False])
def test_sharded_decode_with_a_rank_that_owns_no_tile(fast):
    """Two tiles over three ranks: rank 2 decodes nothing but still takes part in the sequence-parallel estimator / the pooled-statistics
    exchange, and rank 0 still ends with the assembled image."""
    from oracle import vae_oracle as vo
    assert len(vo.split_tiles(40, 30, 16, True)[0]) == 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, fast, q, (40, 30))) for r in range(3)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [f"rank {r}: {msg}" for r, msg in results if msg != "ok"]
    assert not bad, "\n".join(bad)


def _worker_interrupt(rank, world, port, q):
    """Slow mode over two ranks; rank 1 sees state.interrupted in the middle of the lockstep sweep (after its 3rd tile segment).  Both ranks
    must leave the sweep at the SAME pooled barrier (the interrupt rides in the barrier's head exchange), skip the gather together and
    return -- before the fix rank 1's 2-float agreement all-reduce paired with rank 0's head exchange: a spurious NaN error on one side
    and a hang in allreduce_stats on the other."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        torch.set_num_threads(1)
        from hostsim import ldm_decoder as ld
        dec = ld.make_decoder(0, small=True)
        torch.manual_seed(2)
        z = torch.randn(1, 4, 40, 56)
        hook = _hook(dec, 16, True, False)
        hook.shard = (rank, world)
        hook.gather_to = 0
        from modules.shared import state
        state.interrupted = False
        calls = [0]
        orig = hook._run_until_norm

        def counted(steps, st, **kw):
            calls[0] += 1
            if rank == 1 and calls[0] == 3:
                state.interrupted = True
            return orig(steps, st, **kw)

        hook._run_until_norm = counted
        with torch.no_grad():
            out = hook(z)
        assert out.shape == (1, 3, 320, 448) and torch.isfinite(out).all()      # the host's cheap approximation: no tile finished
        assert calls[0] < 40, f"rank {rank} kept decoding after the interrupt ({calls[0]} segments)"
        dist.barrier()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_interrupt_in_multi_rank_slow_mode_leaves_the_sweep_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_interrupt, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [f"rank {r}: {msg}" for r, msg in results if msg != "ok"]
    assert not bad, "\n".join(bad)


# ---- live-window narrowing of the fast-mode decoder tiles (scripts/tilevae.py: live_windows, _live_plan) -----------------------------
def _rec_hook(net, ts, fast=True, rec_convs=True):
    import torch_engine as te
    hook = _hook(net, ts, True, fast)
    # the record-path sweep (_run_tile_rec) on torch doubles; rec_convs=False: convs that the record kernels do not take (exact-fp32 mode,
    # odd channel counts) -- the same sweep on the fp32 hand-over calls, upsample windows through VAEHook._upconv_window_f32
    hook.engine, hook._pack = te.TorchEngineRec(), (te.TorchConvRec if rec_convs else te.TorchConvCounting)
    return hook


@pytest.mark.parametrize("fast", [False, True], ids=["slow_mode", "estimator_pass"])
def test_pooled_statistics_sites_take_the_record_kernels_where_they_pay(monkeypatch, fast):
    """A norm whose statistics are pooled (slow mode; every norm of the fast mode's estimator pass) cannot be applied by the conv that
    produces its input: there the record kernels run behind a conversion pass (rec_from_f32 with the norm's (a, s) + SiLU), and only at the
    sites VAEHook._pooled_site_takes_rec names -- every upsample conv and the 512 -> 512 layers.  Same result as the fp32 hand-over form
    (MDTILE_SLOW_REC=0) and as the oracle."""
    from hostsim import ldm_decoder as ld
    from oracle import vae_oracle as vo
    import torch_engine as te
    torch.manual_seed(5)
    z = torch.randn(1, 4, 36, 44)
    outs, used = {}, {}
    for on in (True, False):
        hook = _rec_hook(ld.make_decoder(0, small=True), 16, fast=fast)
        pl = sys.modules[type(hook).__module__]
        monkeypatch.setattr(pl, "SLOW_REC", on)
        if fast:
            monkeypatch.setattr(pl, "REC_PATH", on)       # (the tile sweep itself off the record doubles: only the estimator's calls are counted)
        te.TorchConvRec.px_computed = 0
        with torch.no_grad():
            outs[on] = hook(z)
        used[on] = te.TorchConvRec.px_computed
        # the rule itself, on stand-in steps: upsample convs always, plain convs only at 512 -> 512
        monkeypatch.setattr(pl, "REC_PATH", True)
        mk = lambda cin, cout, up: pl.Step("conv", conv=type("C", (), {"cin": cin, "cout": cout, "takes_rec": lambda self, u=False: True})(), upsample=up)
        assert hook._pooled_site_takes_rec(mk(256, 256, True)) == on and hook._pooled_site_takes_rec(mk(512, 512, False)) == on
        assert not hook._pooled_site_takes_rec(mk(256, 256, False)) and not hook._pooled_site_takes_rec(mk(512, 256, False))
        assert hook._pooled_site_takes_rec(mk(128, 3, False)) == on      # conv_out behind a pooled norm_out: conversion pass + narrow record conv
    assert used[True] > 0 and used[False] == 0              # the small decoder has no 512-channel layer: its three upsample convs per pass
    assert torch.allclose(outs[True], outs[False], rtol=0, atol=1e-5 * outs[False].abs().max().item())
    with torch.no_grad():
        ref = vo.tiled_forward(ld.make_decoder(0, small=True), z, 16, fast)
    assert (outs[True] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("rec_convs", [True, False], ids=["record_doubles", "handover_doubles"])
def test_slow_mode_takes_the_statistics_the_producing_conv_leaves(monkeypatch, rec_convs):
    """Slow mode (scripts/tilevae.py lockstep loop, upstream :289-361): where the conv that produces a pooled norm's input can leave the
    statistics of its output (PackedConv.leaves_stats -> TileState.stats), GroupNormParam.add_tile takes them instead of a pass over the
    tile (engine.gn_stats).  Same image as with MDTILE_SLOW_STATS=0, same as the oracle; the passes that remain are the ones whose input
    no 3x3 conv with a fused pre-activation produces (conv_in, the attention block's output)."""
    from hostsim import ldm_decoder as ld
    from oracle import vae_oracle as vo
    import torch_engine as te
    torch.manual_seed(9)
    z = torch.randn(1, 4, 36, 44)
    outs, passes, left = {}, {}, {}
    for on in (True, False):
        hook = _rec_hook(ld.make_decoder(0, small=True), 16, fast=False, rec_convs=rec_convs)
        pl = sys.modules[type(hook).__module__]
        monkeypatch.setattr(pl, "SLOW_STATS", on)
        calls = []
        orig = hook.engine.gn_stats
        hook.engine.gn_stats = lambda x, g=32: (calls.append(tuple(x.shape)), orig(x, g))[1]
        te.TorchConv.stats_left = 0
        with torch.no_grad():
            outs[on] = hook(z)
        passes[on], left[on] = len(calls), te.TorchConv.stats_left
    n_tiles = len(vo.split_tiles(36, 44, 16)[0])
    assert left[False] == 0 and left[True] > 0
    assert passes[True] + left[True] == passes[False]              # every (tile, norm) statistic comes from exactly one of the two
    # what is left per tile: conv_in's output and the attention block's (+ the three upsample convs' where only the record kernels leave them)
    assert passes[True] == (2 if rec_convs else 5) * n_tiles and passes[False] == 30 * n_tiles, (passes, n_tiles)
    assert torch.equal(outs[True], outs[False])                    # (the doubles compute both forms with the same torch call)
    with torch.no_grad():
        ref = vo.tiled_forward(ld.make_decoder(0, small=True), z, 16, False)
    assert (outs[True] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


def test_live_windows_of_the_sd_decoder_program():
    """Grow = 1 / 3 / 6 latent px behind the 8x / 4x / 2x upsample convs (3 resblocks per level + conv_out), the 1x level whole; windows
    nest, are clamped to the tile and are given in input px of each upsample conv relative to its already narrowed input plane."""
    from hostsim import ldm_decoder as ld
    hook = _rec_hook(ld.make_decoder(0, small=True), 16)
    pl = sys.modules[type(hook).__module__]
    steps = hook.program()
    ups = [i for i, s in enumerate(steps) if s.kind == "conv" and s.upsample]
    assert len(ups) == 3
    # interior tile: 11 latent px of padding on every side of a 16 x 24 valid rectangle
    win, rect = pl.live_windows(steps, (38, 46), (11, 11, 27, 35))
    assert rect == (10, 10, 28, 36)                                             # valid grown by 1
    assert win[ups[0]] == (5, 5, 28, 36)                                        # 1x plane -> valid grown by 6, in 1x px
    assert win[ups[1]] == (3 * 2, 3 * 2, 22 * 2, 30 * 2)                        # 2x plane (origin 5) -> valid grown by 3: offset 3 latent px
    assert win[ups[2]] == (2 * 4, 2 * 4, 18 * 4, 26 * 4)                        # 4x plane (origin 8) -> valid grown by 1: offset 2 latent px
    # tile in the top-left corner of the image: no padding there, nothing to shed on those sides
    win, rect = pl.live_windows(steps, (27, 35), (0, 0, 16, 24))
    assert rect == (0, 0, 17, 25)
    assert win[ups[0]] == (0, 0, 22, 30) and win[ups[1]] == (0, 0, 19 * 2, 27 * 2) and win[ups[2]] == (0, 0, 17 * 4, 25 * 4)
    # padding smaller than the reach of the convs: only the levels that can shed something get a window
    win, rect = pl.live_windows(steps, (22, 22), (3, 3, 19, 19))
    assert rect == (2, 2, 20, 20) and ups[0] not in win and ups[1] not in win and win[ups[2]] == (2 * 4, 2 * 4, 18 * 4, 18 * 4)
    # the encoder has no upsample conv (and its attention comes last): nothing
    ehook = _hook(ld.make_encoder(0, small=True), 64, False, True)
    assert pl.live_windows(ehook.program(), (128, 128), (32, 32, 96, 96)) == ({}, (0, 0, 128, 128))


@pytest.mark.parametrize("rec_convs", [True, False], ids=["record_convs", "fp32_handover_convs"])
@pytest.mark.parametrize("hw,ts,stacked_origins", [((36, 44), 16, False), ((70, 40), 16, True), ((64, 40), 24, False)])
def test_fast_decode_with_live_windows_matches_oracle_and_the_whole_tile_sweep(hw, ts, stacked_origins, rec_convs):
    """The record-path sweep on torch doubles: narrowed tiles == whole padded tiles == the oracle; and the narrowing does shed work."""
    from hostsim import ldm_decoder as ld
    from oracle import vae_oracle as vo
    import torch_engine as te
    torch.manual_seed(5)
    z = torch.randn(1, 4, *hw)
    with torch.no_grad():
        ref = vo.tiled_forward(ld.make_decoder(0, small=True), z, ts, True)
    outs, px, mixed = {}, {}, {}
    for live in (True, False):
        hook = _rec_hook(ld.make_decoder(0, small=True), ts, rec_convs=rec_convs)
        pl = sys.modules[type(hook).__module__]
        old = pl.LIVE_WINDOW
        pl.LIVE_WINDOW = live
        te.TorchConvRec.px_computed = te.TorchConvRec.window_calls = te.TorchConvRec.mixed_origin_calls = 0
        try:
            with torch.no_grad():
                outs[live] = hook(z)
        finally:
            pl.LIVE_WINDOW = old
        px[live] = te.TorchConvRec.px_computed
        if rec_convs:
            assert (te.TorchConvRec.window_calls > 0) == live
        mixed[live] = te.TorchConvRec.mixed_origin_calls if rec_convs else int(stacked_origins and live)
    # (torch's CPU conv picks its blocking by plane size, so the doubles agree to rounding only; the engine's kernels are bit-identical:
    # tests/test_gpu_rec.py::test_fast_decode_with_live_windows_equals_the_whole_tile_sweep)
    assert (outs[True] - outs[False]).abs().max().item() <= 1e-5 * ref.abs().max().item()      # (exactness: the float64 test below)
    assert (outs[True] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    assert px[True] < 0.9 * px[False], f"narrowing shed only {1 - px[True] / px[False]:.1%} of the conv outputs"
    # (70, 40) at tile 16: the first and the last tile of a column have one shape and windows of one size at different origins
    assert (mixed[True] > 0) == stacked_origins, "stacked sweep of tiles with different window origins (first / last tile of a column)"


def test_live_windows_are_exact_and_tight_in_float64():
    """One tile through the record sweep in float64 (rounding out of the picture): inside the valid rectangle the narrowed sweep equals
    the whole-tile sweep to float64 rounding, and windows one latent pixel smaller do NOT (the bound is the reach of the convs, not slack)."""
    from hostsim import ldm_decoder as ld
    hook = _rec_hook(ld.make_decoder(3, small=True).double(), 16)
    pl = sys.modules[type(hook).__module__]
    steps = hook.program()
    norm_idx = [i for i, s in enumerate(steps) if s.kind == "norm"]
    norm_ord = {i: k for k, i in enumerate(norm_idx)}
    g = torch.Generator().manual_seed(8)
    frozen = [(torch.rand(32, generator=g, dtype=torch.float64) + 0.5, torch.randn(32, generator=g, dtype=torch.float64) * 0.1) for _ in norm_idx]
    E = hook.engine
    coefs = [E.gn_coeffs(m, v, steps[i].norm[0], steps[i].norm[1], steps[i].channels, 32, 1e-6) for i, (v, m) in zip(norm_idx, frozen)]
    th, tw, valid = 38, 40, (11, 11, 27, 29)
    x = torch.randn(1, 4, th, tw, generator=g, dtype=torch.float64)
    with torch.no_grad():
        whole = hook._run_tile_rec(steps, x, frozen, coefs, norm_ord, None)
        windows, rect = pl.live_windows(steps, (th, tw), valid)
        narrowed = hook._run_tile_rec(steps, x, frozen, coefs, norm_ord, windows)
    assert len(windows) == 3 and narrowed.shape[2:] == ((rect[2] - rect[0]) * 8, (rect[3] - rect[1]) * 8)

    def valid_of(t, r):
        return t[:, :, (valid[0] - r[0]) * 8:(valid[2] - r[0]) * 8, (valid[1] - r[1]) * 8:(valid[3] - r[1]) * 8]

    ref = valid_of(whole, (0, 0))
    exact = (valid_of(narrowed, rect) - ref).abs().max().item() / ref.abs().max().item()
    assert exact <= 1e-14, exact                     # rounding of float64 (the doubles' conv blocks by plane size)
    # one latent pixel less at the 2x level (grow 5 = 10 px where 12 are needed): the convs reach the valid rectangle from outside the window
    ups = sorted(windows)
    y0, x0, h, w = windows[ups[0]]
    tight = dict(windows)
    tight[ups[0]] = (y0 + 1, x0 + 1, h - 2, w - 2)
    y0, x0, h, w = windows[ups[1]]
    tight[ups[1]] = (y0 - 2, x0 - 2, h, w)          # same absolute rectangle at the 4x level (its origin is relative to the 2x window)
    with torch.no_grad():
        short = hook._run_tile_rec(steps, x, frozen, coefs, norm_ord, tight)
    assert short.shape == narrowed.shape
    off = (valid_of(short, rect) - ref).abs().max().item() / ref.abs().max().item()
    print(f"narrowed vs whole tile {exact:.1e}; windows one latent px short {off:.1e}")
    assert off > 1e3 * max(exact, 1e-16), (exact, off)      # small (12 random convs damp it) but four orders above rounding
