import os, sys, torch
ROOT = "/root/repo" if os.path.exists("/root/repo/probes") else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")); sys.path.insert(0, ROOT)
import mdtile as E
import _probes_lib
_probes_lib.use(E)      # probe switches exist in the PROBES twin of the library only

dev = torch.device("cuda:0")
def timeit(fn, n=4, rounds=3):
    best = 1e9
    for _ in range(rounds):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best
os.environ["MDTILE_REC_BLOCKS"] = "1"
for cin, cout, H, W in ((128, 128, 2224, 2224), (512, 512, 556, 556)):
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(dev)
    pc = E.PackedConv(conv.weight.detach(), conv.bias.detach())
    x = torch.randn(1, cin, H, W, device=dev)
    ci = torch.stack([torch.rand(1, cin, device=dev) + 0.5, torch.randn(1, cin, device=dev) * 0.3], dim=1).contiguous()
    co = torch.stack([torch.rand(1, cout, device=dev) + 0.5, torch.randn(1, cout, device=dev) * 0.3], dim=1).contiguous()
    xrec = E.rec_from_f32(x, ci)
    rounds = -(-(-(-W // 32) * -(-H // 16)) // 8) * 8 * (cout // 128) / 256
    t_act = timeit(lambda: pc.call_rec(xrec, want_f32=False, want_rec=True, rec_coef=co))
    t_raw = timeit(lambda: pc.call_rec(xrec, want_f32=False, want_rec=True, rec_coef=None))
    t_f32 = timeit(lambda: pc.call_rec(xrec, want_f32=True, want_rec=False))
    os.environ["MDTILE_REC_DBG"] = "1"
    t_k = timeit(lambda: pc.call_rec(xrec, want_f32=False, want_rec=True, rec_coef=co))
    os.environ.pop("MDTILE_REC_DBG")
    print(f"{cin}->{cout} {H}x{W}: per item (us): K loop {t_k*1e3/rounds:6.1f} | rec(act) {t_act*1e3/rounds:6.1f} | rec(raw, no silu) {t_raw*1e3/rounds:6.1f} | fp32 only {t_f32*1e3/rounds:6.1f}", flush=True)
