"""Timing of the tiled VAE ENCODE of one image on the engine (run on the GPU box): python probes/encode_probe.py [side] [tile]
SD-shaped encoder (ch=128, ch_mult 1-2-4-4) with seeded random weights, fast mode; default 8192 x 8192 at encoder tile 3072
(upstream's recommendation for > 16 GB, scripts/tilevae.py:79-87)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
from hostsim import stub_host as sh, ldm_decoder as ld   # stub A1111 host + the random-weight encoder definition (test infra)

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
dev = torch.device("cuda:0")
sh.install(dev)
sh.set_device(dev)
pl = sh.load_plugin()
enc = ld.make_encoder(0).to(dev)
enc.original_forward = enc.forward
hook = pl.tilevae.VAEHook(enc, tile, is_decoder=False, fast_decoder=False, fast_encoder=True, color_fix=False)
x = torch.randn(1, 3, side, side, generator=torch.Generator().manual_seed(1)).to(dev)
import builtins
_p = builtins.print
for it in range(2):
    builtins.print = lambda *a, **k: None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = hook(x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    builtins.print = _p
    print(f"encode {side}x{side} tile {tile}: {dt:.3f} s  -> {tuple(y.shape)}  ({side * side / 64 / dt:.0f} latent-px/s), "
          f"max mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
