"""Timeline of one persistent block of the record conv (csrc/vae_conv_rec.hip, MDTILE_REC_DBG bit 3): s_memtime stamps per wave and item
at item start / K loop done / epilogue code done / vmcnt(0) + barrier of the next item passed -- where the ~12 us between two K loops go."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")); sys.path.insert(0, ROOT)
import mdtile as E
import _probes_lib
_probes_lib.use(E)      # probe switches exist in the PROBES twin of the library only

dev = torch.device("cuda:0")
os.environ["MDTILE_REC_BLOCKS"] = "1"
for cin, cout, H, W in ((128, 128, 2224, 2224), (512, 512, 556, 556)):
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(dev)
    pc = E.PackedConv(conv.weight.detach(), conv.bias.detach())
    x = torch.randn(1, cin, H, W, device=dev)
    res = torch.randn(1, cout, H, W, device=dev)
    ci = torch.stack([torch.rand(1, cin, device=dev) + 0.5, torch.randn(1, cin, device=dev) * 0.3], dim=1).contiguous()
    co = torch.stack([torch.rand(1, cout, device=dev) + 0.5, torch.randn(1, cout, device=dev) * 0.3], dim=1).contiguous()
    xrec = E.rec_from_f32(x, ci)
    forms = {"rec->rec": lambda: pc.call_rec(xrec, want_f32=False, want_rec=True, rec_coef=co),
             "rec->both(+res)": lambda: pc.call_rec(xrec, residual=res, want_f32=True, want_rec=True, rec_coef=co),
             "rec->f32": lambda: pc.call_rec(xrec, want_f32=True),
             "rec->f32+rec": lambda: pc.call_rec(xrec, want_f32=True, want_rec=True, rec_coef=co),
             "rec->f32(+res)": lambda: pc.call_rec(xrec, residual=res, want_f32=True)}
    if "--forms" in sys.argv:
        forms = {k: v for k, v in forms.items() if k in sys.argv[sys.argv.index("--forms") + 1].split(",")}
    XDBG = int(sys.argv[sys.argv.index("--dbg") + 1]) if "--dbg" in sys.argv else 0      # extra MDTILE_REC_DBG bits (16: no fp32 stores, 32: no record stores)
    NARROW = False
    for name, fn in forms.items():
        for _ in range(3):
            fn()
        buf = torch.zeros(64 * 8 * 8, dtype=torch.int64, device=dev)
        os.environ["MDTILE_REC_DBG"] = str((10 if NARROW else 8) | XDBG)
        os.environ["MDTILE_REC_STAMPS"] = hex(buf.data_ptr())
        fn()
        torch.cuda.synchronize()
        os.environ.pop("MDTILE_REC_DBG"); os.environ.pop("MDTILE_REC_STAMPS")
        t = buf.cpu().view(64, 8, 8).double()
        n = int((t[:, 0, 0] > 0).sum().item())
        items = range(2, min(n - 1, 30))       # steady state
        k_loop = torch.stack([t[i, :, 1] - t[i, :, 0] for i in items])                 # [items, waves]
        epi = torch.stack([t[i, :, 2] - t[i, :, 1] for i in items])
        bar = torch.stack([t[i, :, 4] - t[i, :, 2] for i in items])
        period = torch.stack([t[i + 1, :, 0] - t[i, :, 0] for i in items])
        skew_end = torch.stack([t[i, :, 1].max() - t[i, :, 1].min() for i in items])
        f = lambda a: f"{a.mean().item():8.0f} (min {a.min().item():6.0f} max {a.max().item():6.0f})"
        print(f"{cin}->{cout} {H}x{W} {name:16s} [cycles of s_memtime, {n} items stamped] item period {f(period)} | K loop {f(k_loop)} | epilogue code {f(epi)} | "
              f"vmcnt(0) + barrier wait {f(bar)} | spread of the waves' K-loop ends {skew_end.mean().item():6.0f}", flush=True)
