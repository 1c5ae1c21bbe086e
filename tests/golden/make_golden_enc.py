"""Golden vectors for the ENCODER direction of Tiled VAE, produced by the UPSTREAM code (scripts/tilevae.py of the reference
mounted at /root/reference) under hostsim/stub_host.py.  Run here (the reference does not exist on the GPU box):

    python tests/golden/make_golden_enc.py        ->  tests/golden/vae_enc.npz + tests/golden/cases_enc.json
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from hostsim import ldm_decoder as ld, stub_host as sh  # noqa: E402

ENC_CASES = [   # image (not latent) sizes; small encoder (ch=32, same topology as SD's)
    dict(name="enc_fast", H=160, W=192, ts=64, fast=True, color_fix=False, seed=4, enc_seed=0),
    dict(name="enc_slow", H=160, W=192, ts=64, fast=False, color_fix=False, seed=4, enc_seed=0),
    dict(name="enc_color_fix", H=136, W=200, ts=64, fast=True, color_fix=True, seed=5, enc_seed=1),
]


def main():
    sh.install("cpu")
    ref = sh.load_reference()
    out = {}
    for c in ENC_CASES:
        enc = ld.make_encoder(c["enc_seed"], small=True)
        enc.original_forward = enc.forward
        torch.manual_seed(c["seed"])
        x = torch.randn(1, 3, c["H"], c["W"])
        hook = ref.tilevae.VAEHook(enc, c["ts"], is_decoder=False, fast_decoder=False, fast_encoder=c["fast"], color_fix=c["color_fix"])
        y = hook(x)
        out[c["name"] + "/out"] = y.numpy()
        out[c["name"] + "/moments"] = np.array([y.double().sum().item(), (y.double() ** 2).sum().item(), y.abs().max().item()])
    np.savez_compressed(os.path.join(HERE, "vae_enc.npz"), **out)
    with open(os.path.join(HERE, "cases_enc.json"), "w") as f:
        json.dump({"enc": ENC_CASES}, f, indent=1)
    print("vae_enc.npz", os.path.getsize(os.path.join(HERE, "vae_enc.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
