"""NEEDS the instrumented kernel of profiles/r4q/attn_phase_stamps_and_q_dma_ablation_instrumentation.diff (git apply it, rebuild): the product
kernel carries no probing code.  Phases of the attention kernel per key block (csrc/vae_attn_bf16x3.hip, MDTILE_ATTN_STAMPS): s_memtime at block start / scores done /
softmax done / output done, for the 8 waves of block (0, 0); and the same launch with the Q half of the slab DMA left out
(MDTILE_ATTN_DBG=1: wrong numbers) -- is the score phase bound by its L2 -> LDS stream?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")); sys.path.insert(0, ROOT)
import mdtile as E
dev = torch.device("cuda:0")
C = 512
for T in (77284, 30000):
    torch.manual_seed(0)
    q, k, v = torch.randn(1, C, T, device=dev), torch.randn(1, C, T, device=dev), torch.randn(1, C, T, device=dev)
    fn = lambda: E.vae_attn(q, k, v, C ** -0.5, v_channel_major=True)
    for dbg in (0, 1):
        os.environ["MDTILE_ATTN_DBG"] = str(dbg)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); fn(); e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 2
        buf = torch.zeros(64 * 8 * 4, dtype=torch.int64, device=dev)
        os.environ["MDTILE_ATTN_STAMPS"] = hex(buf.data_ptr())
        fn(); torch.cuda.synchronize()
        os.environ.pop("MDTILE_ATTN_STAMPS")
        t = buf.cpu().view(64, 8, 4).double()
        n = int((t[:, 0, 0] > 0).sum().item())
        rng = range(4, min(n - 1, 60))
        f = lambda a: f"{a.mean().item():7.0f} ({a.min().item():6.0f}..{a.max().item():6.0f})"
        score = torch.stack([t[i, :, 1] - t[i, :, 0] for i in rng]); soft = torch.stack([t[i, :, 2] - t[i, :, 1] for i in rng])
        outp = torch.stack([t[i, :, 3] - t[i, :, 2] for i in rng]); per = torch.stack([t[i + 1, :, 0] - t[i, :, 0] for i in rng])
        print(f"T={T} dbg={dbg} ({'Q half of the slabs NOT streamed' if dbg else 'normal'}): {ms:7.2f} ms, {4.0 * T * T * C / ms * 1e-9:6.1f} TF-eq | cycles per key block "
              f"{f(per)} | scores {f(score)} | softmax {f(soft)} | output {f(outp)}   [ideal MFMA issue: 12288 per phase pair-shared SIMD]", flush=True)
    os.environ.pop("MDTILE_ATTN_DBG")
