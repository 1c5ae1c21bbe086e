"""Timing probe for the 1x1 split-bf16 conv (nin_shortcut / q, k, proj_out shapes of one 278x278-latent decoder tile).
python probes/conv1x1_probe.py   (on the GPU box)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
import mdtile as E
if os.environ.get("MDTILE_AB_LIB"):      # another build of the library (same-box A/B of a kernel change)
    E.LIB_PATH = os.environ["MDTILE_AB_LIB"]
elif os.environ.get("MDTILE_C1X1_MT") or os.environ.get("MDTILE_C1X1_STREAM"):      # probe switches: the PROBES twin of the library
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _probes_lib
    _probes_lib.use(E)

dev = torch.device("cuda:0")
torch.manual_seed(0)
BATCH = int(os.environ.get("PROBE_B", "1"))      # 4 = the stacked tiles of the 8K decode
for cin, cout, H, W, res in [(256, 128, 2224, 2224, False), (512, 256, 1112, 1112, False), (512, 512, 278, 278, True), (512, 512, 278, 278, False)]:
    conv = torch.nn.Conv2d(cin, cout, 1).to(dev)
    pc = E.PackedConv(conv.weight.detach(), conv.bias.detach())
    x = torch.randn(BATCH, cin, H, W, device=dev)
    r = torch.randn(BATCH, cout, H, W, device=dev) if res else None
    y = pc(x, residual=r)
    with torch.no_grad():
        ref = torch.nn.functional.conv2d(x[:, :, :64], conv.weight, conv.bias) + (r[:, :, :64] if res else 0)
    err = ((y[:, :, :64] - ref).abs().max() / ref.abs().max()).item()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            pc(x, residual=r)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 5)
    gb = BATCH * (cin + cout * (2 if res else 1)) * H * W * 4 / 1e9
    print(f"1x1 {cin:4d}->{cout:4d} {H}x{W} res={int(res)}: {best:7.3f} ms  {gb / best:6.2f} TB/s  {2.0 * BATCH * H * W * cin * cout / best * 1e-9:6.1f} TF  rel err {err:.1e}", flush=True)
