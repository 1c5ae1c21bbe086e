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
    from oracle import stub_host as sh
    import torch_engine as te
    sh.install("cpu")
    pl = sh.load_plugin()
    net.original_forward = net.forward
    hook = pl.tilevae.VAEHook(net, ts, is_decoder=is_decoder, fast_decoder=fast, fast_encoder=fast, color_fix=color_fix)
    hook.engine, hook._pack, hook._sp_ops = te.TorchEngine(), te.TorchConv, te.TorchSeqParOps()   # torch doubles of the engine
    return hook


@pytest.mark.parametrize("fast", [True, False])
def test_decode_matches_oracle(fast):
    from oracle import ldm_decoder as ld, vae_oracle as vo
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
    from oracle import ldm_decoder as ld, vae_oracle as vo
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
        from oracle import ldm_decoder as ld, vae_oracle as vo
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
        mine = list(range(rank, len(ins), world))
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
