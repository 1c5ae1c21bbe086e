"""Where the record conv's time goes, by switching parts off (run on the GPU box):  python probes/conv_rec_diag.py [--shapes 4,0]
For the one-block-per-CU kernel (csrc/vae_conv_rec.hip) and the two-blocks-per-CU kernel (csrc/vae_conv_rec2.hip): time per item
(= launch time / rounds of items over the resident blocks) of the whole kernel and of the K loop alone (MDTILE_REC_DBG=1: epilogue
skipped), record -> record and record -> fp32 + record with residual; the one-block kernel also on 128 / 64 / 32 CUs only
(MDTILE_REC_GRID: does an item's store epilogue get faster when fewer CUs store at the same time?) and the two-block kernel with ONE
4-wave block per CU (MDTILE_REC2_PER_CU=1: what a 4-wave block does with the CU to itself)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
sys.path.insert(0, ROOT)
import mdtile as E

import _probes_lib
_probes_lib.use(E)      # probe switches exist in the PROBES twin of the library only

dev = torch.device("cuda:0")
SHAPES = [(512, 512, 556, 556), (512, 512, 278, 278), (256, 256, 1112, 1112), (512, 256, 1112, 1112), (128, 128, 2224, 2224),
          (128, 128, 128, 128, 304), (128, 128, 512, 512, 19)]      # 5, 6: the pixel count of 2224^2 as many small images (planes of 64 KB / 1 MB: few pages per item)
if "--shapes" in sys.argv:
    SHAPES = [SHAPES[int(i)] for i in sys.argv[sys.argv.index("--shapes") + 1].split(",")]


def timeit(fn, n=4, rounds=3):
    best = 1e9
    for _ in range(rounds):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


torch.manual_seed(0)
for shape in SHAPES:
    cin, cout, H, W = shape[:4]
    B = shape[4] if len(shape) > 4 else 1
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(dev)
    pc = E.PackedConv(conv.weight.detach(), conv.bias.detach())
    x = torch.randn(B, cin, H, W, device=dev)
    res = torch.randn(B, cout, H, W, device=dev)
    ci = torch.stack([torch.rand(B, cin, device=dev) + 0.5, torch.randn(B, cin, device=dev) * 0.3], dim=1).contiguous()
    co = torch.stack([torch.rand(B, cout, device=dev) + 0.5, torch.randn(B, cout, device=dev) * 0.3], dim=1).contiguous()
    xrec = E.rec_from_f32(x, ci)
    rr = lambda: pc.call_rec(xrec, want_f32=False, want_rec=True, rec_coef=co)
    rb = lambda: pc.call_rec(xrec, residual=res, want_f32=True, want_rec=True, rec_coef=co)
    flops = 2.0 * B * H * W * cout * cin * 9
    ncb = cout // 128
    print(f"{cin}->{cout} {H}x{W} B={B}: per-item times in us (one-block items = 128 couts x 16 rows x 32 px; two-block items are half of that: their times are doubled below)", flush=True)

    def row(label, blocks, rows, nblk, **env):
        base = int(os.environ.get("MDTILE_REC_DBG_BASE", "0"))
        setenv(MDTILE_REC_BLOCKS=blocks, MDTILE_REC_DBG=base or None, **env)
        pt = -(-W // 32) * -(-H // rows)
        items = -(-pt // 8) * 8 * ncb * B
        rounds = -(-items // nblk)
        scale = 16 // rows       # per 16-row item equivalent
        out = []
        for fn in (rr, rb):
            t_full = timeit(fn)
            setenv(MDTILE_REC_DBG=1 | base)
            t_k = timeit(fn)
            setenv(MDTILE_REC_DBG=base or None)
            out.append((t_full, t_k))
        (a, ak), (b, bk) = out
        per = lambda t: t * 1e3 / rounds * scale * (nblk / (256 * (2 if rows == 8 else 1))) if False else t * 1e3 / rounds * scale
        print(f"   {label:34s} rec->rec {a:7.3f} ms {flops / a * 1e-9:6.1f} TF  item {per(a):6.1f} | K loop only {ak:7.3f} ms {flops / ak * 1e-9:6.1f} TF item {per(ak):6.1f}"
              f" || rec->both {b:7.3f} ms {flops / b * 1e-9:6.1f} TF item {per(b):6.1f} | K loop only {bk:7.3f} ms item {per(bk):6.1f}", flush=True)
        for k in env:
            setenv(**{k: None})

    row("one block / CU, 256 CUs", 1, 16, 256)
    for g in (128, 64, 32):
        row(f"one block / CU, {g} CUs", 1, 16, g, MDTILE_REC_GRID=g)
    row("two blocks / CU (512 blocks)", 2, 8, 512)
    row("two-block kernel, skew off", 2, 8, 512, MDTILE_REC2_SKEW=0)
    row("two-block kernel, ONE block / CU", 2, 8, 256, MDTILE_REC2_PER_CU=1)
    row("two-block kernel, 1 block/CU, 64 CUs", 2, 8, 64, MDTILE_REC2_PER_CU=1, MDTILE_REC_GRID=64)
    del x, res, xrec
