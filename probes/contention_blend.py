"""The blend kernels (csrc/blend.hip) sharing the GPU with other processes' MFMA kernels: is every launch the uncontended launch?
(Round 6: conv_in's round-5 form dropped products of `v_pk_fma_f32 ... op_sel:[0,1,0]` under exactly this load, probes/contention_fewcin.py;
blend.hip's MoD path compiles to `v_pk_mul_f32 ... op_sel:[0,1]`, so it gets the same test.  The blend is bit-exact by contract.)
    python probes/contention_blend.py [K] [R] [mix:<text>|same:<text>]"""
import os, sys
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "probes")); sys.path.insert(0, ROOT)
from contention_fewcin import background, setup      # noqa: E402

CASES = [      # W, H, tile, overlap, method, regions
    ("cfg1 md 1024^2 tile128 ov8", 1024, 1024, 128, 8, "md", ()),
    ("mod 1024^2 tile128 ov8", 1024, 1024, 128, 8, "mod", ()),
    ("cfg3 mod 512^2 tile96 ov48 (LDS-staged kernel)", 512, 512, 96, 48, "mod", ()),
    ("cfg4 md 1024^2 tile128 ov64", 1024, 1024, 128, 64, "md", ()),
    ("cfg5 mod + regions", 512, 512, 96, 48, "mod", ((0.1, 0.1, 0.5, 0.5, "bg", 0.0), (0.4, 0.3, 0.4, 0.6, "fg", 0.2))),
]

if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    mode = sys.argv[3] if len(sys.argv) > 3 else "mix:handover"
    ctx = mp.get_context("spawn")
    E = setup()
    import bench
    dev = torch.device("cuda:0")
    built = []
    for name, W, H, tile, ov, meth, regions in CASES:
        plan, gather, blend, nbytes = bench.blend_setup(E, dev, W, H, tile, ov, 8, meth, regions)
        blend(); torch.cuda.synchronize()
        ref = blend.out.clone()
        blend(); torch.cuda.synchronize()
        print(f"alone, {name}: deterministic {bool(torch.equal(ref, blend.out))}", flush=True)
        built.append((name, blend, ref))
    side, ops = None, None
    if mode.startswith("same"):
        import contention_ops
        ops = {k: f for k, f in contention_ops.make_ops(E, dev).items() if mode.split(":", 1)[1] in k}
        side = torch.cuda.Stream()
        ps = []
    else:
        stop = ctx.Event()
        readies = [ctx.Event() for _ in range(K - 1)]
        ps = [ctx.Process(target=background, args=(mode, stop, r)) for r in readies]
        for p in ps:
            p.start()
        for r in readies:
            r.wait(300)
    for name, blend, ref in built:
        nbad, nel = 0, 0
        for r in range(R):
            if side is not None and r % 20 == 0:
                with torch.cuda.stream(side):
                    for f in ops.values():
                        f()
            blend.out.fill_(float("nan"))
            blend()
            if not torch.equal(blend.out, ref):
                nbad += 1
                nel += int((blend.out != ref).sum().item())
        if side is not None:
            side.synchronize()
        print(f"{mode}, {name}: {nbad} of {R} contended launches differ from the uncontended one ({nel} elements)", flush=True)
    if ps:
        stop.set()
        for p in ps:
            p.join(60)
