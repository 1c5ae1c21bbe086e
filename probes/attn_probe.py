"""Timing probe for the attention kernels (run on the GPU box): python probes/attn_probe.py [T ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
import mdtile as E
if os.environ.get("MDTILE_AB_LIB"):      # another build of the library (counter passes of a kernel variant: tools/gpu_r6.sh sqattn)
    E.LIB_PATH = os.environ["MDTILE_AB_LIB"]

dev = torch.device("cuda:0")
ZEROS = "--zeros" in sys.argv     # all-zero q / k / v: same instruction stream, no operand toggling (DVFS / power check)
QUICK = "--quick" in sys.argv     # one mid-size problem, split-bf16 kernel only (counter passes)
EXACT_ONLY = "--exact-only" in sys.argv     # the exact-fp32 kernel only (counter passes of k_attn)
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
        if (exact and (T > 40000 or QUICK)) or (EXACT_ONLY and not exact):
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


if "--ab-zeros" in sys.argv:
    # same process, alternating random / all-zero q, k, v (identical instruction stream): how much of the time is the clock
    T = Ts[-1]
    torch.manual_seed(1)
    qr, kr, vr = torch.randn(1, C, T, device=dev), torch.randn(1, C, T, device=dev), torch.randn(1, T, C, device=dev)
    qz, kz, vz = torch.zeros_like(qr), torch.zeros_like(kr), torch.zeros_like(vr)
    flops = 4.0 * T * T * C
    for rep in range(3):
        for name, (a, b, c) in (("random", (qr, kr, vr)), ("zeros ", (qz, kz, vz))):
            E.vae_attn(a, b, c, C ** -0.5)
            torch.cuda.synchronize()
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(4):
                E.vae_attn(a, b, c, C ** -0.5)
            e_.record()
            torch.cuda.synchronize()
            ms = s_.elapsed_time(e_) / 4
            print(f"  T={T} {name}: {ms:8.3f} ms {flops / ms * 1e-9:7.1f} TF", flush=True)
