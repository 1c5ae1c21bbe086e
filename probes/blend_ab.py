"""A/B of the blend kernel on the 8K grid (1024x1024 latent, 81 tiles 128x128, overlap 8, N=2, C=4): (planes per thread, candidates per chunk)
configurations (nontemporal tile loads / canvas stores were A/B'd here in round 2 and measured slower for the default (8,2):
17.1 -> 17.8 us; the option is gone).  python probes/blend_ab.py   (on the GPU box)"""
import os, sys, torch
sys.path.insert(0, "multidiffusion-upscaler-for-automatic1111_amd"); sys.path.insert(0, ".")
import mdtile as E
import _probes_lib
_probes_lib.use(E)      # probe switches exist in the PROBES twin of the library only

dev = torch.device("cuda:0")
W = H = 1024; tw = th = 128; ov = 8; N, C = 2, 4
plan = E.Plan(W, H, tw, th, ov, 8)
T = plan.num_tiles
packed = torch.randn(T * N, C, th, tw, device=dev)
weights = torch.zeros(H, W, device=dev)
E.weight_map_add_grid(plan, None, weights)
out = torch.empty(N, C, H, W, device=dev)
def run(tag, n=50):
    call = E.BlendCall(plan, E.METHOD_MD, [packed], N, C, weights=weights, out=out, packed=True)
    for _ in range(5): call()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): call()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    print(f"{tag}: {best:.2f} us  {80216064 / best * 1e-3:.0f} GB/s", flush=True)
os.environ["MDTILE_BLEND_CFG"] = "0,0"
for n in (1, 5, 20, 50, 200):
    run(f"default cfg, {n:3d} launches per event pair", n)
if "--data" in sys.argv:
    # operand dependence (the HBM path has a data-dependent cost too?): zeros / random tiles
    packed.zero_(); run("zero tiles, 50 launches"); packed.normal_(); run("random tiles, 50 launches")
for cfg in ("0,0", "8,2", "8,4", "4,2", "4,4", "2,4"):
    os.environ["MDTILE_BLEND_CFG"] = cfg
    run(f"cfg {cfg}")
# (round 3: a residency throttle -- fewer resident blocks per CU through an unused dynamic LDS allocation, so that late blocks load while early
# ones store -- was A/B'd here and measured slower in every configuration, profiles/r3b/blend_ab_residency_throttle.log; the knob is gone)
