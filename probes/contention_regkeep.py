"""Do waves get their registers back when several PROCESSES share one GPU?  (probes/csrc/regkeep_probe.hip; round 6)
K - 1 background processes run the op mix of probes/contention_ops.py; the foreground runs the register-persistence kernel R times for
each register count and prints every mismatch it records: (block, wave, lanes, VGPR index, xor).
    python probes/contention_regkeep.py [K] [R] [mix|torch|none]"""
import collections, ctypes, os, sys
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "probes"))
from contention_fewcin import background, setup      # noqa: E402


def run(lib, nregs, spin, grid, R, dev):
    cap = 1 << 16
    out = torch.zeros(cap, 4, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    total, hits = 0, collections.Counter()
    bad_calls = 0
    for _ in range(R):
        cnt.zero_()
        rc = lib.mdtile_probe_regkeep(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), ctypes.c_uint(cap), nregs, spin, grid,
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        n = int(cnt.item())
        if n:
            bad_calls += 1
            total += n
            rec = out[:min(n, cap)].cpu().numpy().astype("uint32")
            for blk, thr, reg, x in rec:
                hits[(int(thr) & 63) // 16, int(reg)] += 1
    return bad_calls, total, hits


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    mode = sys.argv[3] if len(sys.argv) > 3 else "mix"
    ctx = mp.get_context("spawn")
    E = setup()
    import _probes_lib
    _probes_lib.use(E)
    lib = E.lib()
    dev = torch.device("cuda:0")
    for nregs in (0, 32, 100, 240):
        b, t, h = run(lib, nregs, 2000 if nregs else 40, 2048, 5, dev)
        print(f"alone, {nregs} VGPRs: {b} of 5 calls with mismatches ({t} registers x lanes)", flush=True)
    if mode != "none":
        stop = ctx.Event()
        readies = [ctx.Event() for _ in range(K - 1)]
        ps = [ctx.Process(target=background, args=(mode, stop, r)) for r in readies]
        for p in ps:
            p.start()
        for r in readies:
            r.wait(300)
        for nregs in (0, 32, 100, 240):
            b, t, h = run(lib, nregs, 2000 if nregs else 40, 2048, R, dev)
            print(f"{K - 1} background processes ({mode}), {nregs if nregs else 'LDS broadcast reads, 0'} VGPRs: {b} of {R} calls with mismatches ({t} registers x lanes)", flush=True)
            if h:
                q = collections.Counter()
                regs = collections.Counter()
                for (quarter, reg), n in h.items():
                    q[quarter] += n; regs[reg] += n
                print(f"    by lane quarter (0: lanes 0-15 .. 3: lanes 48-63): {dict(sorted(q.items()))}")
                print(f"    by VGPR index (top 12): {regs.most_common(12)}")
        stop.set()
        for p in ps:
            p.join(60)
