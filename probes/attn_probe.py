"""Timing probe for the attention kernels (run on the GPU box): python probes/attn_probe.py [T ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
import mdtile as E

dev = torch.device("cuda:0")
ZEROS = "--zeros" in sys.argv     # all-zero q / k / v: same instruction stream, no operand toggling (DVFS / power check)
QUICK = "--quick" in sys.argv     # one mid-size problem, split-bf16 kernel only (counter passes)
Ts = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or ([30000] if QUICK else [7396, 30000, 71168, 77284])
C = 512
for T in Ts:
    torch.manual_seed(0)
    q, k = torch.randn(1, C, T, device=dev), torch.randn(1, C, T, device=dev)
    v = torch.randn(1, T, C, device=dev)
    if ZEROS:
        q.zero_(), k.zero_(), v.zero_()
    scale = C ** -0.5
    flops = 4.0 * T * T * C
    line = f"T={T:6d} C={C}: "
    outs = {}
    for exact in (False, True):
        if exact and (T > 40000 or QUICK):
            continue
        o = E.vae_attn(q, k, v, scale, exact=exact)
        torch.cuda.synchronize()
        n = 2
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            o = E.vae_attn(q, k, v, scale, exact=exact)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        outs[exact] = o
        line += f"{'f32   ' if exact else 'bf16x3'} {ms:9.3f} ms {flops / ms * 1e-9:7.1f} TF   "
    if len(outs) == 2:
        line += f"max dev {((outs[False] - outs[True]).abs().max() / outs[True].abs().max()).item():.2e}"
    print(line, flush=True)
