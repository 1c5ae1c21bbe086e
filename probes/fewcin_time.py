"""conv_in timing, shipping vs a probe form (MDTILE_FEWCIN_FORM, PROBES twin): decoder 4 -> 512 on a 278^2 latent tile, encoder 3 -> 128 on a 3072^2 image tile."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")); sys.path.insert(0, os.path.join(ROOT, "probes"))
import mdtile as E
if os.environ.get("MDTILE_FEWCIN_FORM"):
    import _probes_lib
    _probes_lib.use(E)
dev = torch.device("cuda:0")
torch.manual_seed(0)
for cin, cout, hw in ((4, 512, 278), (3, 128, 3072), (3, 128, 1200)):
    c = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    pc = E.PackedConv(c.weight.detach(), c.bias.detach())
    z = torch.randn(1, cin, hw, hw, device=dev)
    y = pc(z)
    with torch.no_grad():
        want = torch.nn.functional.conv2d(z, c.weight, c.bias, padding=1)
    err = (y - want).abs().max().item()
    for _ in range(5):
        pc(z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        pc(z)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    gb = (cout + cin) * hw * hw * 4 / 1e9
    print(f"form {os.environ.get('MDTILE_FEWCIN_FORM', 'shipping')}: {cin}->{cout} {hw}^2: {us:8.1f} us  {gb / us * 1e6:7.0f} GB/s  max|y - torch| {err:.2e}", flush=True)
