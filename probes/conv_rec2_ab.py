"""A/B of the record conv kernels: ONE 8-wave block per CU (csrc/vae_conv_rec.hip, MDTILE_REC_BLOCKS=1) against TWO independent
4-wave blocks per CU (csrc/vae_conv_rec2.hip, default), in one process on one GPU:  python probes/conv_rec2_ab.py [--shapes 0,2] [--census]
For every shape: rec -> rec, rec -> fp32 (+ residual), rec -> both, each timed under both kernels (and, for the two-block form, with
the start-up skew off / by block index / by the per-CU arrival counter, and the skew length varied); the outputs of the two
forms are compared bit for bit (same per-accumulator MFMA order).  --census prints where the blocks of a two-block launch ran."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
sys.path.insert(0, ROOT)
import mdtile as E

import _probes_lib
_probes_lib.use(E)      # probe switches exist in the PROBES twin of the library only

dev = torch.device("cuda:0")
SHAPES = [  # cin, cout, H, W (output), upsample
    (512, 512, 556, 556, False),
    (512, 512, 1112, 1112, True),
    (256, 256, 1112, 1112, False),
    (256, 256, 2224, 2224, True),
    (128, 128, 2224, 2224, False),
    (512, 512, 278, 278, False),
    (512, 512, 556, 556, True),
    (512, 256, 1112, 1112, False),
    (256, 128, 2224, 2224, False),
    (512, 512, 278, 278, False, 2),      # 9 .. 12: the 1x level of a decoder tile, stacked 2 / 3 deep (TILE_BATCH), and its un-padded form
    (512, 512, 278, 278, False, 3),
    (512, 512, 256, 256, False),
    (512, 512, 556, 556, True, 3),
    (512, 512, 139, 139, False),         # 13: half-size tile (decoder tile 128)
    (512, 512, 86, 86, False, 3),        # 14: decoder tile 64, three tiles
]
if "--shapes" in sys.argv:
    SHAPES = [SHAPES[int(i)] for i in sys.argv[sys.argv.index("--shapes") + 1].split(",")]
SWEEP = "--sweep" in sys.argv
ZEROS = "--zeros" in sys.argv      # all-zero activations AND weights: same instruction stream, no operand toggling (DVFS / power check)


def timeit(fn, n=5, rounds=3):
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


def census(fn, grid=512):
    buf = torch.zeros(4096, dtype=torch.int32, device=dev)
    setenv(MDTILE_REC2_CENSUS=hex(buf.data_ptr()))
    fn()
    torch.cuda.synchronize()
    setenv(MDTILE_REC2_CENSUS=None)
    v = buf.cpu().tolist()[:grid]
    keys = [x & 0x7FFFFFFF for x in v]
    second = [(x >> 31) & 1 for x in v]
    from collections import Counter
    per_cu = Counter(keys)
    hist = Counter(per_cu.values())
    pairs_idx = sum(1 for b in range(grid // 2) if keys[b] == keys[b + grid // 2])
    both = Counter()
    for k, s in zip(keys, second):
        both[k] += s
    return (f"{len(per_cu)} distinct CU ids, blocks per id {dict(hist)}, XCDs {sorted({k >> 8 for k in keys})}, "
            f"(b, b + grid/2) on the same CU: {pairs_idx}/{grid // 2}, ids with exactly one delayed block: {sum(1 for k in both if both[k] == 1)}/{len(per_cu)}")


torch.manual_seed(0)
for shape in SHAPES:
    cin, cout, H, W, up = shape[:5]
    B = shape[5] if len(shape) > 5 else 1
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(dev)
    if ZEROS:
        with torch.no_grad():
            conv.weight.zero_()
    pc = E.PackedConv(conv.weight.detach(), conv.bias.detach())
    hin, win = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(B, cin, hin, win, device=dev)
    if ZEROS:
        x.zero_()
    res = None if up else torch.randn(B, cout, H, W, device=dev)
    coef_in = torch.stack([torch.rand(B, cin, device=dev) + 0.5, torch.randn(B, cin, device=dev) * 0.3], dim=1).contiguous()
    coef_out = torch.stack([torch.rand(B, cout, device=dev) + 0.5, torch.randn(B, cout, device=dev) * 0.3], dim=1).contiguous()
    if ZEROS:
        coef_in.zero_()
    xrec = E.rec_from_f32(x, None if up else coef_in)
    flops = 2.0 * B * H * W * cout * cin * (4 if up else 9)        # EXECUTED flops (the sub-pixel form runs 4 taps)
    forms = {
        "rec->rec ": lambda: pc.call_rec(xrec, upsample2x=up, want_f32=False, want_rec=True, rec_coef=coef_out),
        "rec->both": lambda: pc.call_rec(xrec, residual=res, upsample2x=up, want_f32=True, want_rec=True, rec_coef=coef_out),
    }
    print(f"{cin:4d}->{cout:4d} {H}x{W}{' up' if up else '   '} B={B}", flush=True)
    for name, fn in forms.items():
        setenv(MDTILE_REC_BLOCKS=1)
        y1, r1 = fn()
        t1 = timeit(fn)
        setenv(MDTILE_REC_BLOCKS=2, MDTILE_REC2_SKEW=None, MDTILE_REC2_SKEW_PCT=None)
        y2, r2 = fn()
        same = (y1 is None or torch.equal(y1, y2)) and torch.equal(r1.records(), r2.records())
        t2 = timeit(fn)
        line = f"   {name}: one block {t1:7.3f} ms {flops / t1 * 1e-9:6.1f} TF | two blocks {t2:7.3f} ms {flops / t2 * 1e-9:6.1f} TF ({(t1 / t2 - 1) * 100:+5.1f} %) bit-identical {same}"
        if SWEEP:
            for skew, pct in ((0, 100), (1, 100), (2, 50), (2, 150), (2, 200)):
                setenv(MDTILE_REC2_SKEW=skew, MDTILE_REC2_SKEW_PCT=pct)
                t = timeit(fn)
                line += f" | skew {skew}@{pct}: {t:7.3f}"
            setenv(MDTILE_REC2_SKEW=None, MDTILE_REC2_SKEW_PCT=None)
        print(line, flush=True)
    if "--census" in sys.argv:
        for skew in (2, 1):
            setenv(MDTILE_REC_BLOCKS=2, MDTILE_REC2_SKEW=skew)
            print(f"   census (skew mode {skew}): " + census(forms["rec->rec "]), flush=True)
        setenv(MDTILE_REC2_SKEW=None)
    del x, res, xrec
