"""Same-process A/B of the record 3x3 conv kernels (run on the GPU box): ONE 8-wave block per CU with the store epilogue behind
each item's K loop (csrc/vae_conv_rec.hip, MDTILE_CONV_REC_ONE_BLOCK) against the DRIPPED epilogue (probes/csrc/vae_conv_recd.hip,
MDTILE_CONV_REC_DRIP: 64-cout items, previous item's stores / next item's residual issued in slots between the K-steps).
    python probes/conv_drip_ab.py [--shapes 0,3] [--b 1]
Prints TFLOP/s-equivalent (2 * B * H * W * cout * cin * 9) per output kind; both families are bit-identical (tests/test_gpu_rec.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
sys.path.insert(0, ROOT)
import mdtile as E
DBG = None
if "--kloop" in sys.argv:      # K loops alone (MDTILE_REC_DBG=1 of the PROBES twin: no epilogue / no slots, nothing is written)
    DBG = "1"
if "--dbg" in sys.argv:        # probes/csrc/vae_conv_recd.hip: 2 no activation arithmetic | 4 no record stores | 8 no fp32 stores | 16 no residual loads
    DBG = sys.argv[sys.argv.index("--dbg") + 1]
if DBG is not None:
    sys.path.insert(0, os.path.join(ROOT, "probes"))
    import _probes_lib
    _probes_lib.use(E)
    os.environ["MDTILE_REC_DBG"] = DBG

dev = torch.device("cuda:0")
SHAPES = [  # cin, cout, H, W
    (128, 128, 2224, 2224),
    (256, 128, 2224, 2224),
    (256, 256, 1112, 1112),
    (512, 256, 1112, 1112),
    (512, 512, 556, 556),
    (512, 512, 278, 278),
]
if "--shapes" in sys.argv:
    SHAPES = [SHAPES[int(i)] for i in sys.argv[sys.argv.index("--shapes") + 1].split(",")]
B = int(sys.argv[sys.argv.index("--b") + 1]) if "--b" in sys.argv else 1


def timeit(fn, n=5):
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
for cin, cout, H, W in SHAPES:
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(dev)
    pc = E.PackedConv(conv.weight.detach(), conv.bias.detach())
    x = torch.randn(B, cin, H, W, device=dev)
    res = torch.randn(B, cout, H, W, device=dev)
    coef_in = torch.stack([torch.rand(B, cin, device=dev) + 0.5, torch.randn(B, cin, device=dev) * 0.3], dim=1).contiguous()
    coef_out = torch.stack([torch.rand(B, cout, device=dev) + 0.5, torch.randn(B, cout, device=dev) * 0.3], dim=1).contiguous()
    xrec = E.rec_from_f32(x, coef_in)
    del x
    flops = 2.0 * B * H * W * cout * cin * 9
    line = f"{cin:4d}->{cout:4d} {H}x{W} B={B}: "
    kinds = [("rec->rec (conv1)", dict(want_f32=False, want_rec=True, rec_coef=coef_out)),
             ("rec->f32+rec+res (conv2)", dict(residual=res, want_f32=True, want_rec=True, rec_coef=coef_out)),
             ("rec->f32+res", dict(residual=res, want_f32=True, want_rec=False))]
    for name, kw in kinds:
        ts = {}
        for rnd in range(2):          # alternate the families: same clocks, same neighbours
            for fam, flag in (("one", E.CONV_REC_ONE_BLOCK), ("drip", E.CONV_REC_DRIP)):
                t = timeit(lambda: pc.call_rec(xrec, family=flag, **kw))
                ts[fam] = min(ts.get(fam, 1e9), t)
        line += f"| {name}: one {flops / ts['one'] * 1e-9:6.1f} drip {flops / ts['drip'] * 1e-9:6.1f} TF ({(ts['one'] / ts['drip'] - 1) * 100:+5.1f} %) "
    print(line, flush=True)
    del xrec, res
    torch.cuda.empty_cache()
