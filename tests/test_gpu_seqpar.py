"""GPU leg of the sequence-parallel estimator: the row-band primitives through the C ABI (mdtile_gn_sums /
mdtile_gn_from_sums, mdtile_vae_attn_qk) and the whole `estimate_group_norm_sp` on the engine with TWO processes that share
cuda:0 (gloo moves the halos / statistics / keys+values through the host -- the data path is the product's, only the
transport differs from RCCL), against the single-process estimator of the VAEHook."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hostsim import ldm_decoder as ld

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN = os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")


def test_gn_row_band_sums(plugin, cuda):
    E = plugin.engine
    torch.manual_seed(0)
    x = torch.randn(2, 64, 11, 13) * 2.0 + 0.5
    xg = x.to(cuda)
    parts = [(0, 4), (4, 5), (5, 11)]
    sums = sum(E.gn_sums(xg, lo, hi, 32) for lo, hi in parts)
    var, mean = E.gn_from_sums(sums, float((64 // 32) * 11 * 13))
    v_ref, m_ref = E.gn_stats(xg, 32)
    assert torch.allclose(mean, m_ref, rtol=1e-6, atol=1e-7) and torch.allclose(var, v_ref, rtol=1e-6, atol=1e-7)
    ref = x.double().view(2 * 32, -1)
    assert torch.allclose(mean.cpu().double(), ref.mean(1), rtol=1e-6, atol=1e-7)
    assert torch.allclose(var.cpu().double(), ref.var(1, unbiased=False), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("B,C,Tq,Tk", [(1, 512, 300, 1100), (2, 128, 64, 200), (1, 256, 1000, 130), (1, 512, 2100, 2100)])
def test_attention_band_queries_all_keys(plugin, cuda, B, C, Tq, Tk):
    E = plugin.engine
    torch.manual_seed(Tq + Tk)
    q, k, v = torch.randn(B, C, Tq), torch.randn(B, C, Tk) * 1.5, torch.randn(B, Tk, C)
    scale = float(int(C) ** (-0.5))
    w = torch.softmax(torch.bmm(q.permute(0, 2, 1), k) * scale, dim=2)
    ref = torch.bmm(w, v).permute(0, 2, 1)
    out = E.vae_attn_qk(q.to(cuda), k.to(cuda), v.to(cuda), scale).cpu()
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    assert err < 5e-5, f"band attention rel err {err}"


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, H, W, q):
    try:
        for p in (ROOT, PLUGIN):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from hostsim import stub_host as sh
        dev = torch.device("cuda:0")
        sh.install(dev)
        sh.set_device(dev)
        pl = sh.load_plugin()
        from mdtile import seqpar
        dec = ld.make_decoder(3).to(dev)                       # real SD widths (attention at C = 512)
        dec.original_forward = dec.forward
        hook = pl.tilevae.VAEHook(dec, 64, is_decoder=True, fast_decoder=True, fast_encoder=False, color_fix=False)
        steps = hook.program()
        torch.manual_seed(11)
        zs = torch.randn(1, 4, H, W).to(dev)
        with torch.no_grad():
            ref = hook.estimate_group_norm(zs, steps)
            got = seqpar.estimate_group_norm_sp(steps, zs, seqpar.BandComm(rank, world), seqpar.EngineOps(), pl.tilevae.FUSE_PRE_GN)
        assert len(got) == len(ref) == 30
        for i, ((v, m), (vr, mr)) in enumerate(zip(got, ref)):
            assert torch.allclose(m, mr, rtol=3e-4, atol=3e-5), f"rank {rank} norm {i}: mean differs by {(m - mr).abs().max().item()}"
            assert torch.allclose(v, vr, rtol=3e-4, atol=3e-5), f"rank {rank} norm {i}: var differs by {(v - vr).abs().max().item()}"
        dist.barrier()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W", [(2, 22, 18), (3, 25, 16)])
def test_sequence_parallel_estimator_on_engine(plugin, cuda, world, H, W):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    bad = [f"rank {r}: {msg}" for r, msg in results if msg != "ok"]
    assert not bad, "\n".join(bad)
