"""Round-6 A/B of the blend, one process, 20 back-to-back launches per HIP-event pair, median of 7 rounds, COLD (8 rotating buffer sets,
> the 256 MiB Infinity Cache) and WARM (one static set), on four grids: 8K 128 / 8 (the headline), 8K 128 / 64, cfg3 4096^2 96 / 48
Mixture-of-Diffusers, cfg2 2048^2 96 / 48:
    default            what mdtile_blend dispatches: k_blend_lds where it applies (csrc/blend.hip)
    lds SQ,LPP         the LDS-staged kernel in other shapes (strip quads, planes per block; MDTILE_BLEND_LDS, PROBES twin)
    k_blend PP,G       the register-path kernel (MDTILE_BLEND_LDS=0) in its (planes per thread, candidates per chunk) shapes
    copy               mdtile_stream_copy of the same number of bytes: the floor of one launch of that size
every output is compared bit for bit with the default's.  MDTILE_AB_LIB=<libmdtile.so of another build> times that build's default only
(e.g. the round-5 kernel: no write-through stores, no non-temporal loads, no LDS form).
    python probes/blend_r6_ab.py            (on the GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mdtile as E          # noqa: E402
import _probes_lib          # noqa: E402

OTHER = os.environ.get("MDTILE_AB_LIB", "")
if OTHER:
    E.LIB_PATH = OTHER
    print(f"[ab] timing the default dispatch of {OTHER}", flush=True)
else:
    _probes_lib.use(E)
dev = torch.device("cuda:0")


def timed(calls, n=20, rounds=7):
    for c in calls[:3]:
        c()
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(n):
            calls[i % len(calls)]()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / n * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def report(name, us, nbytes):
    med, best = us
    print(f"  {name:34s} {med:7.2f} us (best {best:6.2f})  {nbytes / med * 1e-3:7.0f} GB/s  {nbytes / med * 1e-3 / 8000:.3f} of 8 TB/s", flush=True)


GRIDS = ((1024, 1024, 128, 8, "md"), (1024, 1024, 128, 64, "md"), (512, 512, 96, 48, "mod"), (256, 256, 96, 48, "md"))
for (W, H, tile, ov, m) in GRIDS:
    N, C = 2, 4
    method = E.METHOD_MD if m == "md" else E.METHOD_MOD
    plan = E.Plan(W, H, tile, tile, ov, 4)
    weights = torch.zeros(1, 1, H, W, device=dev)
    tile_w = E.gaussian_weights(plan.tile_w, plan.tile_h, dev) if method == E.METHOD_MOD else None
    E.weight_map_add_grid(plan, tile_w, weights)
    rescale = E.reciprocal(weights) if method == E.METHOD_MOD else None
    kw = dict(weights=weights) if method == E.METHOD_MD else dict(tile_w=tile_w, rescale=rescale)
    nbytes = 4 * (plan.num_tiles * N * C * plan.tile_h * plan.tile_w + N * C * H * W) + 4 * H * W * (1 if method == E.METHOD_MD else 2)
    sets = max(8, int(700e6 // nbytes) + 1)
    bufs = [(torch.randn(plan.num_tiles * N, C, plan.tile_h, plan.tile_w, device=dev), torch.empty(N, C, H, W, device=dev)) for _ in range(sets)]
    print(f"{W}x{H} latent, {plan.num_tiles} tiles {plan.tile_w}x{plan.tile_h} overlap {plan.overlap}, {m}: {nbytes / 1e6:.1f} MB algorithmic per launch, {sets} rotating sets when cold")

    def calls(sel):
        return [E.BlendCall(plan, method, [t], N, C, out=o, packed=True, **kw) for t, o in sel]

    variants = [("default dispatch", None)]
    if not OTHER:
        variants += [(f"lds SQ,LPP = {v}", ("MDTILE_BLEND_LDS", v)) for v in ("256,4", "256,2", "128,4", "128,2", "64,4")]
        variants += [(f"k_blend PP,G = {v}", ("MDTILE_BLEND_CFG", v)) for v in ("8,2", "8,4", "4,2", "2,4")]
    t0, o0 = bufs[0]
    os.environ.pop("MDTILE_BLEND_LDS", None); os.environ.pop("MDTILE_BLEND_CFG", None)
    ref = torch.empty_like(o0)
    E.BlendCall(plan, method, [t0], N, C, out=ref, packed=True, **kw)()
    torch.cuda.synchronize()

    def setenv(v):
        os.environ.pop("MDTILE_BLEND_LDS", None); os.environ.pop("MDTILE_BLEND_CFG", None)
        if v is not None:
            os.environ[v[0]] = v[1]
            if v[0] == "MDTILE_BLEND_CFG":
                os.environ["MDTILE_BLEND_LDS"] = "0"

    for name, v in variants[1:]:
        setenv(v)
        o0.fill_(float("nan"))
        calls(bufs[:1])[0]()
        torch.cuda.synchronize()
        if not torch.equal(o0, ref):
            print(f"  {name}: NOT bit-identical to the default dispatch (max abs diff {(o0 - ref).abs().max().item()})")
    for label, sel in (("COLD", bufs), ("WARM (one static set)", bufs[:1])):
        print(f" {label}")
        for name, v in variants:
            setenv(v)
            report(name, timed(calls(sel)), nbytes)
        setenv(None)
        half = (nbytes // 2 + 4095) // 4096 * 4096
        cps = [E.StreamCopyCall(torch.randn(half // 4, device=dev), torch.empty(half // 4, device=dev)) for _ in range(len(sel))]
        report(f"stream copy {half / 1e6:.1f} MB -> {half / 1e6:.1f} MB", timed(cps), 2 * half)
        del cps
    del bufs
    torch.cuda.empty_cache()
