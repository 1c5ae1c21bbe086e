"""Timing probe for the 3x3 conv kernels (run on the GPU box): python probes/conv_probe.py [--shapes 0,2,5] [--exact]
Prints TFLOP/s-equivalent (2*B*H*W*cout*cin*9; the sub-pixel upsample form EXECUTES 4/9 of that) for the decoder's conv
shapes of one 278x278-latent tile: record-image kernels (vae_conv_rec.hip: fp32 output / activated record output / both) against
the fp32 hand-over kernels (vae_conv_bf16x3.hip, plain and with the fused GroupNorm+SiLU staging), and their max deviation."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
sys.path.insert(0, ROOT)
import mdtile as E

import _probes_lib
_probes_lib.use(E)      # probe switches exist in the PROBES twin of the library only

dev = torch.device("cuda:0")
SHAPES = [  # cin, cout, H, W, upsample
    (512, 512, 278, 278, False),
    (512, 512, 556, 556, True),
    (512, 512, 556, 556, False),
    (512, 512, 1112, 1112, True),
    (512, 256, 1112, 1112, False),
    (256, 256, 1112, 1112, False),
    (256, 256, 2224, 2224, True),
    (256, 128, 2224, 2224, False),
    (128, 128, 2224, 2224, False),
]
if "--shapes" in sys.argv:
    SHAPES = [SHAPES[int(i)] for i in sys.argv[sys.argv.index("--shapes") + 1].split(",")]
EXACT = "--exact" in sys.argv
ZEROS = "--zeros" in sys.argv      # all-zero activations AND weights: same instruction stream, no operand toggling (DVFS / power check)


def timeit(fn, n=6):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


torch.manual_seed(0)
for cin, cout, H, W, up in SHAPES:
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(dev)
    if ZEROS:
        with torch.no_grad():
            conv.weight.zero_()
    pc = E.PackedConv(conv.weight.detach(), conv.bias.detach())
    hin, win = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(1, cin, hin, win, device=dev)
    if ZEROS:
        x.zero_()
    res = torch.randn(1, cout, H, W, device=dev)
    coef_in = torch.stack([torch.rand(1, cin, device=dev) + 0.5, torch.randn(1, cin, device=dev) * 0.3], dim=1).contiguous()
    if ZEROS:
        coef_in.zero_()         # silu(0 x + 0) = 0: the staged operands are exact zeros too
    coef_out = torch.stack([torch.rand(1, cout, device=dev) + 0.5, torch.randn(1, cout, device=dev) * 0.3], dim=1).contiguous()
    flops = 2.0 * H * W * cout * cin * 9
    line = f"{cin:4d}->{cout:4d} {H}x{W}{' up' if up else '   '}: "
    t_old = timeit(lambda: pc(x, residual=res, upsample2x=up))
    line += f"f32-in {t_old:7.3f} ms {flops / t_old * 1e-9:6.1f} TF | "
    if not up:
        t_gn = timeit(lambda: pc(x, residual=res, pre_gn=coef_in))
        line += f"f32-in+GN {t_gn:7.3f} ms {flops / t_gn * 1e-9:6.1f} TF | "
    xrec = E.rec_from_f32(x, None if up else coef_in)
    t_prep = timeit(lambda: E.rec_from_f32(x, None if up else coef_in))
    t_r32 = timeit(lambda: pc.call_rec(xrec, residual=res, upsample2x=up, want_f32=True))
    t_rrec = timeit(lambda: pc.call_rec(xrec, upsample2x=up, want_f32=False, want_rec=True, rec_coef=coef_out))
    t_both = timeit(lambda: pc.call_rec(xrec, residual=res, upsample2x=up, want_f32=True, want_rec=True, rec_coef=coef_out))
    os.environ["MDTILE_REC_PERSIST"] = "0"      # A/B in the same process: one item per block instead of the persistent grid
    t_rrec1 = timeit(lambda: pc.call_rec(xrec, upsample2x=up, want_f32=False, want_rec=True, rec_coef=coef_out))
    t_both1 = timeit(lambda: pc.call_rec(xrec, residual=res, upsample2x=up, want_f32=True, want_rec=True, rec_coef=coef_out))
    os.environ["MDTILE_REC_PERSIST"] = "1"
    t_rrec2 = timeit(lambda: pc.call_rec(xrec, upsample2x=up, want_f32=False, want_rec=True, rec_coef=coef_out))
    line += (f"rec->f32 {t_r32:7.3f} ms {flops / t_r32 * 1e-9:6.1f} TF | rec->rec {t_rrec:7.3f} ms {flops / t_rrec * 1e-9:6.1f} TF | "
             f"rec->both {t_both:7.3f} ms {flops / t_both * 1e-9:6.1f} TF | prep {t_prep:6.3f} ms"
             f" | 1-item/block: rec->rec {t_rrec1:7.3f} both {t_both1:7.3f} | persistent again rec->rec {t_rrec2:7.3f}")
    ya = pc(x, residual=res, upsample2x=up, pre_gn=None if up else coef_in)
    yb, _ = pc.call_rec(xrec, residual=res, upsample2x=up, want_f32=True)
    line += f" | dev(rec, f32-in) {((ya - yb).abs().max() / ya.abs().max()).item():.1e}"
    if EXACT:
        xa = x if up else torch.nn.functional.silu(x * coef_in[:, 0].view(1, -1, 1, 1) + coef_in[:, 1].view(1, -1, 1, 1))
        ye = pc(xa, residual=res, upsample2x=up, exact=True)
        line += f" dev(rec, exact) {((ye - yb).abs().max() / ye.abs().max()).item():.1e}"
    print(line, flush=True)
    del x, res, ya, yb, xrec

if "--fit" in sys.argv:
    # per-item cost model of the record conv: same image and cout, K depth varied -> time per item = a * NK + b
    print("per-item model (1112x1112, cout 128, rec->rec): items = pixel tiles (35 x 70 = 2450) x cout blocks", flush=True)
    pts = []
    for cin in (128, 256, 512):
        conv = torch.nn.Conv2d(cin, 128, 3, 1, 1).to(dev)
        pc = E.PackedConv(conv.weight.detach(), conv.bias.detach())
        x = torch.randn(1, cin, 1112, 1112, device=dev)
        ci = torch.stack([torch.rand(1, cin, device=dev) + 0.5, torch.randn(1, cin, device=dev) * 0.3], dim=1).contiguous()
        co = torch.stack([torch.rand(1, 128, device=dev) + 0.5, torch.randn(1, 128, device=dev) * 0.3], dim=1).contiguous()
        xrec = E.rec_from_f32(x, ci)
        res = torch.randn(1, 128, 1112, 1112, device=dev)
        t_rr = min(timeit(lambda: pc.call_rec(xrec, want_f32=False, want_rec=True, rec_coef=co)) for _ in range(3))
        t_both = min(timeit(lambda: pc.call_rec(xrec, residual=res, want_f32=True, want_rec=True, rec_coef=co)) for _ in range(3))
        items = 35 * 70
        rounds = -(-items // 256)
        pts.append((cin // 16, t_rr * 1e3 / rounds, t_both * 1e3 / rounds))
        print(f"  cin {cin:4d} (NK {cin // 16:2d}): rec->rec {t_rr:7.3f} ms = {t_rr * 1e3 / rounds:7.2f} us per item-round | rec->both {t_both:7.3f} ms = {t_both * 1e3 / rounds:7.2f} us", flush=True)
        del x, xrec, res
    (n0, a0, b0), (n1, a1, b1), (n2, a2, b2) = pts
    slope = (a2 - a0) / (n2 - n0)
    print(f"  fit rec->rec : {slope:6.3f} us per K-step + {a0 - slope * n0:6.2f} us fixed per item   (ideal MFMA time per K-step: 5.76 us @ 2.4 GHz, 7.28 us @ 1.9 GHz)")
    slope_b = (b2 - b0) / (n2 - n0)
    print(f"  fit rec->both: {slope_b:6.3f} us per K-step + {b0 - slope_b * n0:6.2f} us fixed per item")
