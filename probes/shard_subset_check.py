"""Does a rank's SUBSET of the tiles decode to the same pixels as the full single-rank sweep?  One process, no collectives: fast mode with the
replicated estimator (MDTILE_SP_ESTIMATOR=0), VAEHook.shard = (r, world), gather_to = None -- every 'rank' r is run in turn and its tiles'
rectangles are compared with the plain decode.   python probes/shard_subset_check.py [latent] [world] [repeats]"""
import os, sys
os.environ["MDTILE_SP_ESTIMATOR"] = "0"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
from hostsim import stub_host as sh, ldm_decoder as ld
dev = torch.device("cuda:0")
sh.install(dev); sh.set_device(dev)
pl = sh.load_plugin()
from mdtile import sharding
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dec = ld.make_decoder(0).to(dev); dec.original_forward = dec.forward
z = torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(2)).to(dev)
import builtins
_p = builtins.print
def quiet(f):
    builtins.print = lambda *a, **k: None
    try: return f()
    finally: builtins.print = _p
solo = pl.tilevae.VAEHook(dec, 256, True, True, False, False)
ref = quiet(lambda: solo(z)).float()
den = ref.abs().max().item()
ins, outs = solo.split_tiles(L, L)
owner = sharding.deal_tiles(ins, world)
for rep in range(reps):
    for r in range(world):
        hook = pl.tilevae.VAEHook(dec, 256, True, True, False, False)
        hook.shard = (r, world)
        img = quiet(lambda: hook(z)).float()
        errs = [(i, float('%.1e' % ((img[:, :, ob[2]:ob[3], ob[0]:ob[1]] - ref[:, :, ob[2]:ob[3], ob[0]:ob[1]]).abs().max().item() / den))) for i, ob in enumerate(outs) if owner[i] == r]
        _p(f"rep {rep} rank {r}/{world}: {errs}", flush=True)
