"""Timing probe for the conv kernels (run on the GPU box): python probes/conv_probe.py
Prints TFLOP/s (fp32-equivalent: 2*B*H*W*cout*cin*k*k) for the decoder's conv shapes, split-bf16 vs exact-fp32, and the
max relative deviation between the two."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
sys.path.insert(0, ROOT)
import mdtile as E

dev = torch.device("cuda:0")
SHAPES = [  # cin, cout, k, H, W, upsample   (one 278x278-latent decoder tile of the 8K decode)
    (512, 512, 3, 278, 278, False),
    (512, 512, 3, 556, 556, True),
    (512, 512, 3, 556, 556, False),
    (512, 512, 3, 1112, 1112, True),
    (512, 256, 3, 1112, 1112, False),
    (256, 256, 3, 1112, 1112, False),
    (256, 256, 3, 2224, 2224, True),
    (256, 128, 3, 2224, 2224, False),
    (128, 128, 3, 2224, 2224, False),
]
# usage: conv_probe.py [--no-exact] [--shapes 0,2,5]     (env MDTILE_CONV_TH=16 / MDTILE_UPCONV=direct select kernel variants)
NO_EXACT = "--no-exact" in sys.argv
if "--shapes" in sys.argv:
    SHAPES = [SHAPES[int(i)] for i in sys.argv[sys.argv.index("--shapes") + 1].split(",")]
print(f"variant: MDTILE_CONV_TH={os.environ.get('MDTILE_CONV_TH', '8')} MDTILE_UPCONV={os.environ.get('MDTILE_UPCONV', 'subpixel')}", flush=True)
torch.manual_seed(0)
for cin, cout, k, H, W, up in SHAPES:
    conv = torch.nn.Conv2d(cin, cout, k, 1, k // 2).to(dev)
    pc = E.PackedConv(conv.weight.detach(), conv.bias.detach())
    hin, win = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(1, cin, hin, win, device=dev)
    res = torch.randn(1, cout, H, W, device=dev)
    flops = 2.0 * H * W * cout * cin * k * k
    outs = {}
    line = f"{cin:4d}->{cout:4d} k{k} {H}x{W}{' up' if up else '   '}: "
    for exact in ((False,) if NO_EXACT else (False, True)):
        y = pc(x, residual=res, upsample2x=up, exact=exact)
        torch.cuda.synchronize()
        n = 3 if exact else 6
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            y = pc(x, residual=res, upsample2x=up, exact=exact)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        outs[exact] = y
        line += f"{'f32   ' if exact else 'bf16x3'} {ms:8.3f} ms {flops / ms * 1e-9:7.1f} TF   "
    err = float("nan") if NO_EXACT else ((outs[False] - outs[True]).abs().max() / outs[True].abs().max()).item()
    print(line + f"max dev {err:.2e}", flush=True)
    del x, res, outs, y
