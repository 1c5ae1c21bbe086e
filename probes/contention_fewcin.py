"""conv_in (k_conv3x3_fewcin) under GPU contention from other PROCESSES: is its result the uncontended one, and is either of them right?
K - 1 background processes keep the GPU busy (mode 'mdtile': the same conv; mode 'torch': a torch matmul loop); the foreground process
compares R contended calls with its uncontended reference and with torch's conv2d.
    python probes/contention_fewcin.py [K] [R] [mdtile|torch]       MDTILE_FEWCIN_FORM=1: the round-5 kernel (PROBES twin)"""
import os, sys, time
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def setup():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")); sys.path.insert(0, os.path.join(ROOT, "probes"))
    import mdtile as E
    if os.environ.get("MDTILE_FEWCIN_FORM"):
        import _probes_lib
        _probes_lib.use(E)
    return E


def background(mode, stop, ready):
    E = setup()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    if mode.startswith("mix"):      # the op mix of probes/contention_ops.py (long conv / attention kernels); "mix:<text>": only the ops whose name contains <text>
        import contention_ops
        ops = contention_ops.make_ops(E, dev)
        if ":" in mode:
            ops = {k: f for k, f in ops.items() if mode.split(":", 1)[1] in k}
            assert ops, mode
        def fn():
            for f in ops.values():
                f()
    elif mode == "torch":
        a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
        fn = lambda: a @ b
    else:
        c = torch.nn.Conv2d(4, 512, 3, padding=1)
        pc = E.PackedConv(c.weight.detach().to(dev), c.bias.detach().to(dev))
        z = torch.randn(2, 4, 278, 278, device=dev)
        fn = lambda: pc(z)
    fn(); torch.cuda.synchronize()
    ready.set()
    while not stop.is_set():
        for _ in range(1 if mode.startswith("mix") else 20):
            fn()
        torch.cuda.synchronize()


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    mode = sys.argv[3] if len(sys.argv) > 3 else "mdtile"
    ctx = mp.get_context("spawn")
    E = setup()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    c = torch.nn.Conv2d(4, 512, 3, padding=1).to(dev)
    pc = E.PackedConv(c.weight.detach(), c.bias.detach())
    z = torch.randn(2, 4, 278, 278, device=dev)
    with torch.no_grad():
        want = torch.nn.functional.conv2d(z.double(), c.weight.double(), c.bias.double(), padding=1)
    ref = pc(z).clone()
    again = pc(z)
    torch.cuda.synchronize()
    print(f"alone: deterministic {bool(torch.equal(ref, again))}, max |y - fp64 conv| {(ref.double() - want).abs().max().item():.2e}", flush=True)
    if mode.startswith("same"):      # "same:<text>": no other process -- the ops whose name contains <text> run on a SIDE STREAM of this process
        import contention_ops
        ops = {k: f for k, f in contention_ops.make_ops(E, dev).items() if mode.split(":", 1)[1] in k}
        side = torch.cuda.Stream()
        nbad, lanes = 0, set()
        for r in range(R):
            with torch.cuda.stream(side):
                for f in ops.values():
                    f()
            y = pc(z)
            side.synchronize()
            if not torch.equal(y, ref):
                nbad += 1
                lanes |= set((((y != ref).nonzero()[:, 3] % 128) // 2).unique().tolist())
        print(f"same process, {len(ops)} ops on a side stream ({mode}): {nbad} of {R} overlapped calls differ; lanes hit {sorted(lanes)}", flush=True)
        sys.exit(0)
    stop = ctx.Event()
    readies = [ctx.Event() for _ in range(K - 1)]
    ps = [ctx.Process(target=background, args=(mode, stop, r)) for r in readies]
    for p in ps:
        p.start()
    for r in readies:
        r.wait(300)
    nbad, worst, lanes = 0, 0.0, set()
    for r in range(R):
        y = pc(z)
        if not torch.equal(y, ref):
            nbad += 1
            worst = max(worst, (y.double() - want).abs().max().item())
            idx = (y != ref).nonzero()
            lanes |= set(((idx[:, 3] % 128) // 2).unique().tolist())
            if nbad <= 2:      # the shape of one failure: which (batch, row, 128-px block) groups, and how many of the 512 couts in each
                grp = {}
                for b_, co_, r_, x_ in idx.tolist():
                    grp.setdefault((b_, r_, x_ // 128), set()).add(co_)
                desc = ", ".join(f"(b{b_} row {r_} xblock {xb_}): {len(cs)} couts [{min(cs)}..{max(cs)}]" for (b_, r_, xb_), cs in sorted(grp.items())[:6])
                print(f"    failure {nbad}: {len(grp)} (batch, row, block) groups: {desc}", flush=True)
                if nbad == 1 and os.environ.get("FEWCIN_DUMP"):      # everything an offline analysis needs (tools: numpy on the CPU box)
                    import numpy as np
                    k = idx[:200000]
                    np.savez_compressed(os.environ["FEWCIN_DUMP"], z=z.cpu().numpy(), w=c.weight.detach().cpu().numpy(), bias=c.bias.detach().cpu().numpy(), idx=k.cpu().numpy(),
                                        got=y[k[:, 0], k[:, 1], k[:, 2], k[:, 3]].cpu().numpy(), ref=ref[k[:, 0], k[:, 1], k[:, 2], k[:, 3]].cpu().numpy())
                # is a wrong value the conv of ANOTHER cout's weights (a stale LDS address in those lanes) with this cout's bias?
                bias = c.bias.detach()
                hist = {}
                for b_, co_, r_, x_ in idx[:4000].tolist():
                    cand = ref[b_, :, r_, x_] - bias + bias[co_]
                    d = (cand - y[b_, co_, r_, x_]).abs()
                    j = int(d.argmin())
                    key = (j - co_) if float(d[j]) < 2e-6 else "none"
                    hist[key] = hist.get(key, 0) + 1
                print(f"    wrong value == conv with the weights of cout (co + d), histogram of d over {min(len(idx), 4000)} wrong elements: {dict(sorted(hist.items(), key=lambda kv: -kv[1])[:8])}", flush=True)
    stop.set()
    for p in ps:
        p.join(60)
    tag = "round-5 form (one copy of the weights, op_sel broadcasts)" if os.environ.get("MDTILE_FEWCIN_FORM") == "1" else "shipping (weights as ready-made pairs, no op_sel)"
    print(f"{tag}, {K - 1} background processes ({mode}): {nbad} of {R} contended calls differ from the uncontended result; worst |y - fp64 conv| {worst:.2e}; lanes hit {sorted(lanes)}", flush=True)
