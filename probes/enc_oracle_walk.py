"""Which op of the ORACLE-on-GPU path (oracle/gpu_reference.py: torch fp32 with banded convs) goes wrong on a 3072^2 encoder image?  Walks the
first ops of the encoder like oracle/vae_oracle.py: estimate_stats and checks a few output rows of every conv against an fp64 conv of the same rows."""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
from hostsim import ldm_decoder as ld
from oracle import gpu_reference as gr, vae_oracle as vo
import torch.nn.functional as F
dev = torch.device("cuda:0")
enc = ld.make_encoder(0).to(dev)
for S in (2048, 3072):
    x = torch.randn(1, 3, S, S, generator=torch.Generator().manual_seed(1)).to(dev)
    ops = vo.build_ops(enc, False)
    tile, res = x, []
    with torch.no_grad(), gr.reference_arithmetic():
        for i, (kind, mod) in enumerate(ops[:14]):
            prev = tile
            if kind == "norm":
                var, mean = vo.get_var_mean(tile, 32)
                tile = vo.custom_group_norm(tile, 32, mean, var, mod.weight, mod.bias)
                chk = ""
                # direct check of the statistics and of one row
                v4 = prev.view(1, 32, -1)
                m64, var64 = v4.double().mean(dim=2)[0], v4.double().var(dim=2, unbiased=False)[0]
                chk = f"var_mean vs fp64: mean {float((mean.double() - m64).abs().max()):.2e} var rel {float(((var.double() - var64).abs() / var64).max()):.2e}"
            elif kind == "store_res":
                res.append(tile if mod is None else mod(tile)); chk = ""
            elif kind == "add_res":
                tile = tile + res.pop(); chk = ""
            elif kind == "silu":
                tile = F.silu(tile); chk = f"silu max err {float((tile[:, :, 1000:1002] - F.silu(prev[:, :, 1000:1002].double()).float()).abs().max()):.2e}"
            else:
                tile = mod(tile)
                chk = ""
                if isinstance(mod, torch.nn.Conv2d) and mod.kernel_size == (3, 3) and mod.stride == (1, 1):
                    errs = []
                    for r0 in (0, 73, 74, 75, 76, 1000, tile.shape[2] - 3):
                        lo, hi = max(0, r0 - 1), min(prev.shape[2], r0 + 3)
                        xb = F.pad(prev[:, :, lo:hi].double(), (1, 1, 1 - (r0 - lo), 1 - (hi - (r0 + 2))))
                        ref = gr._orig_conv2d(xb, mod.weight.double(), mod.bias.double(), 1, 0, 1, 1)
                        errs.append(float((tile[:, :, r0:r0 + 2].double() - ref).abs().max() / ref.abs().max()))
                    chk = "rows vs fp64: " + " ".join(f"{e:.1e}" for e in errs)
            print(f"S={S} op {i:2d} {kind:9s} {tuple(tile.shape)}  {chk}", flush=True)
    del x, tile, res, prev
    torch.cuda.empty_cache()
