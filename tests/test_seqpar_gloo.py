"""Sequence-parallel fast-mode estimator (mdtile/seqpar.py) on CPU: world_size 1, 2 and 3 over gloo.

The product runs `estimate_group_norm_sp` with `EngineOps` (mdtile C-ABI calls, GPU only).  Here the TEST injects plain
torch ops with the same interface so that the host logic -- row partition, halo-slot protocol around 3x3 / upsample convs,
fp64 statistics all-reduce, key/value all-gather, the Tq != Tk attention call -- is checked without a GPU against an
unsplit run of the same network: the frozen (var, mean) of all 30 GroupNorms must agree on every rank.
(The GPU leg, tests/test_gpu_seqpar.py, runs the same executor on the engine with two processes sharing cuda:0.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN = os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")


class TorchOps:
    """torch restatement of the ops interface of mdtile/seqpar.py (test double for EngineOps)."""

    def ksize(self, conv):
        return conv.kernel_size[0]

    def fuses_pre_gn(self, conv, upsample):
        return conv.kernel_size[0] == 3 and not upsample and conv.in_channels % 16 == 0 and conv.out_channels >= 32

    def conv(self, conv, x, residual=None, upsample2x=False, pre_gn=None, token_major=False):
        if pre_gn is not None:
            a, s = pre_gn[:, 0], pre_gn[:, 1]
            x = F.silu(x * a[:, :, None, None] + s[:, :, None, None])
        if upsample2x:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        y = conv(x)
        if residual is not None:
            y = y + residual
        if token_major:
            B, C, H, W = y.shape
            y = y.permute(0, 2, 3, 1).reshape(B, H * W, C).contiguous()
        return y

    def tanh(self, x):
        return torch.tanh(x)

    def gn_sums(self, x, row_lo, row_hi):
        B, C = x.shape[:2]
        v = x[:, :, row_lo:row_hi, :].double().reshape(B * 32, -1)
        return torch.stack([v.sum(1), (v * v).sum(1)], dim=1)

    def gn_from_sums(self, sums, count):
        m = sums[:, 0] / count
        v = (sums[:, 1] / count - m * m).clamp_min(0.0)
        return v.float(), m.float()

    def gn_coeffs(self, mean, var, gamma, beta, C):
        B = mean.numel() // 32
        cpg = C // 32
        rstd = 1.0 / torch.sqrt(var.view(B, 32, 1) + 1e-6)
        a = (rstd * gamma.view(1, 32, cpg)).reshape(B, C)
        s = (beta.view(1, 32, cpg) - mean.view(B, 32, 1) * a.view(B, 32, cpg)).reshape(B, C)
        return torch.stack([a, s], dim=1)

    def gn_apply(self, x, mean, var, gamma, beta, silu, inplace):
        c = self.gn_coeffs(mean, var, gamma, beta, x.shape[1])
        y = x * c[:, 0, :, None, None] + c[:, 1, :, None, None]
        return F.silu(y) if silu else y

    def attn_qk(self, q, k, v_tok, scale):
        w = torch.softmax(torch.bmm(q.permute(0, 2, 1), k) * scale, dim=2)      # [B, Tq, Tk]
        return torch.bmm(w, v_tok).permute(0, 2, 1).contiguous()                # [B, C, Tq]


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _program():
    from hostsim import ldm_decoder as ld, stub_host as sh
    sh.install("cpu")
    pl = sh.load_plugin()
    dec = ld.make_decoder(4, small=True)
    with torch.no_grad():
        steps = pl.tilevae.build_task_queue(dec, True, pack=lambda conv: conv)
    return steps


def _reference(steps, zs):
    """Unsplit estimator with the same torch ops (world 1 through the same executor) AND an independent plain walk."""
    from mdtile import seqpar
    return seqpar.estimate_group_norm_sp(steps, zs, seqpar.BandComm(0, 1), TorchOps(), fuse_pre_gn=True)


def _worker(rank, world, port, H, W, fuse, q):
    try:
        for p in (ROOT, PLUGIN):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(1)
        from mdtile import seqpar
        steps = _program()
        torch.manual_seed(3)
        zs = torch.randn(1, 4, H, W)
        with torch.no_grad():
            ref = _reference(steps, zs)
            got = seqpar.estimate_group_norm_sp(steps, zs, seqpar.BandComm(rank, world), TorchOps(), fuse_pre_gn=fuse)
        assert len(got) == len(ref) == 30
        for i, ((v, m), (vr, mr)) in enumerate(zip(got, ref)):
            assert torch.allclose(m, mr, rtol=2e-4, atol=2e-5), f"rank {rank} norm {i}: mean differs by {(m - mr).abs().max()}"
            assert torch.allclose(v, vr, rtol=2e-4, atol=2e-5), f"rank {rank} norm {i}: var differs by {(v - vr).abs().max()}"
        dist.barrier()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [f"rank {r}: {msg}" for r, msg in results if msg != "ok"]
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("world,H,W,fuse", [(2, 12, 10, True), (3, 13, 8, True), (3, 9, 11, False)])
def test_sequence_parallel_estimator_matches_unsplit(world, H, W, fuse):
    _run(world, H, W, fuse)


def test_unsplit_executor_matches_oracle_estimator():
    """world = 1 through the band executor == the oracle's estimate_stats (upstream tilevae.py:464-505 restated)."""
    for p in (ROOT, PLUGIN):
        if p not in sys.path:
            sys.path.insert(0, p)
    from hostsim import ldm_decoder as ld
    from oracle import vae_oracle as vo
    steps = _program()
    torch.manual_seed(3)
    zs = torch.randn(1, 4, 12, 10)
    with torch.no_grad():
        got = _reference(steps, zs)
        ref = vo.estimate_stats(vo.build_ops(ld.make_decoder(4, small=True)), zs)
    assert len(got) == len(ref) == 30
    for k, ((v, m), (vr, mr)) in enumerate(zip(got, ref)):
        assert torch.allclose(m, mr, rtol=2e-4, atol=2e-5) and torch.allclose(v, vr, rtol=2e-4, atol=2e-5), f"norm {k}"
