"""A/B of the start-up stagger of the persistent record-conv blocks (csrc/vae_conv_rec.hip: stagger_start; MDTILE_REC_STAGGER_PCT is read per
launch): the decoder's conv shapes x output forms, spread = 0 / 25 / 50 / 75 / 100 % of an estimated item period.  ms per launch, median of 7."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")); sys.path.insert(0, ROOT)
import mdtile as E

import _probes_lib
_probes_lib.use(E)      # probe switches exist in the PROBES twin of the library only

dev = torch.device("cuda:0")
os.environ["MDTILE_REC_BLOCKS"] = "1"
SHAPES = [(128, 128, 2224, 2224, False, 1), (256, 128, 2224, 2224, False, 1), (256, 256, 1112, 1112, False, 1), (512, 512, 556, 556, False, 1),
          (512, 512, 278, 278, False, 3), (256, 256, 2224, 2224, True, 1), (512, 512, 1112, 1112, True, 1)]
PCTS = [int(v) for v in sys.argv[sys.argv.index("--pcts") + 1].split(",")] if "--pcts" in sys.argv else [0, 25, 50, 75, 100]


def med(fn, n=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts)


torch.manual_seed(0)
for cin, cout, H, W, up, B in SHAPES:
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(dev)
    pc = E.PackedConv(conv.weight.detach(), conv.bias.detach())
    hin, win = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(B, cin, hin, win, device=dev)
    res = torch.randn(B, cout, H, W, device=dev)
    ci = torch.stack([torch.rand(B, cin, device=dev) + 0.5, torch.randn(B, cin, device=dev) * 0.3], dim=1).contiguous()
    co = torch.stack([torch.rand(B, cout, device=dev) + 0.5, torch.randn(B, cout, device=dev) * 0.3], dim=1).contiguous()
    xrec = E.rec_from_f32(x, None if up else ci)
    forms = {"rec->rec": lambda: pc.call_rec(xrec, upsample2x=up, want_f32=False, want_rec=True, rec_coef=co),
             "rec->f32+rec": lambda: pc.call_rec(xrec, upsample2x=up, want_f32=True, want_rec=True, rec_coef=co)}
    if not up:
        forms["rec->f32+rec (+res)"] = lambda: pc.call_rec(xrec, residual=res, want_f32=True, want_rec=True, rec_coef=co)
    ref = {}
    for name, fn in forms.items():
        line = f"{cin:4d}->{cout:4d} {H}x{W}{' up' if up else '   '} x{B} {name:20s}"
        for pct in PCTS:
            os.environ["MDTILE_REC_STAGGER_PCT"] = str(pct)
            t = med(fn)
            if pct == PCTS[0]:
                ref[name] = t
            line += f" | {pct:3d}%: {t:7.3f} ms ({(t / ref[name] - 1) * 100:+5.1f}%)"
        print(line, flush=True)
os.environ.pop("MDTILE_REC_STAGGER_PCT", None)
