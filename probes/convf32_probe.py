"""The exact-fp32 3x3 conv (csrc/vae_conv.hip: k_conv, v_mfma_f32_32x32x2_f32) at the decoder's three wide shapes: TFLOP/s against the 157 TF peak.
MDTILE_CONVF32_FORM (PROBES twin) forces a block shape: 0 = 8-channel slabs, one block per CU; 1 = 4-channel slabs, two blocks; 3 = 2-channel slabs, two blocks."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")); sys.path.insert(0, os.path.join(ROOT, "probes"))
import mdtile as E
if os.environ.get("MDTILE_CONVF32_FORM"):
    import _probes_lib
    _probes_lib.use(E)
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, cin, cout, hw in ((1, 512, 512, 278), (4, 512, 512, 278), (1, 256, 256, 556), (1, 128, 128, 1112), (1, 512, 256, 556)):
    c = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    pc = E.PackedConv(c.weight.detach(), c.bias.detach())
    z = torch.randn(B, cin, hw, hw, device=dev)
    y = pc(z, exact=True)
    with torch.no_grad():
        want = torch.nn.functional.conv2d(z.double(), c.weight.double(), c.bias.double(), padding=1)
    err = (y.double() - want).abs().max().item() / want.abs().max().item()
    for _ in range(3):
        pc(z, exact=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        pc(z, exact=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = 2 * 9 * B * cin * cout * hw * hw / ms / 1e9
    print(f"form {os.environ.get('MDTILE_CONVF32_FORM', 'shipping dispatch')}: {B} x {cin}->{cout} {hw}^2: {ms:7.3f} ms  {tf:6.1f} TF  {tf / 157.3:.3f} of 157.3  rel err vs fp64 {err:.1e}", flush=True)
