"""Is every kernel of the decode correct when OTHER processes load the same GPU?  (Round 6: the N-rank bench flow run as N processes on one
device -- `bench.py --gpus N --debug-single-device` -- showed single tiles 1-3 % off at full size, while any subset of tiles decoded
alone in one process is bit-identical: a hand-counted LDS-DMA wait that only holds while the chip is otherwise idle would look like this.)
K worker processes share cuda:0.  Phase 1, one after the other: each decodes its latent alone -> reference.  Phase 2, all at once: each
decodes R more times and compares every tile rectangle with its reference.  Any difference is a race (the decode is deterministic).
    python probes/contention_determinism.py [K] [R] [latent] [vae_tile]         env: MDTILE_REC=0, MDTILE_ATTN_MODE=f32, MDTILE_CONV_MODE=f32, ...
"""
import os, sys, time
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(k, K, R, L, ts, turn, go, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
    import builtins
    from hostsim import stub_host as sh, ldm_decoder as ld
    dev = torch.device("cuda:0")
    sh.install(dev); sh.set_device(dev)
    pl = sh.load_plugin()
    _p = builtins.print
    builtins.print = lambda *a, **kw: None
    dec = ld.make_decoder(0).to(dev); dec.original_forward = dec.forward
    z = torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(2 + k)).to(dev)
    hook = pl.tilevae.VAEHook(dec, ts, True, True, False, False)
    ins, outs = hook.split_tiles(L, L)
    while turn.value != k:
        time.sleep(0.01)
    ref = hook(z).float()
    ref2 = hook(z).float()
    alone = bool(torch.equal(ref, ref2))
    torch.cuda.synchronize()
    with turn.get_lock():
        turn.value += 1
    go.wait()
    den = ref.abs().max().item()
    bad = []
    for r in range(R):
        img = hook(z).float()
        for i, ob in enumerate(outs):
            e = (img[:, :, ob[2]:ob[3], ob[0]:ob[1]] - ref[:, :, ob[2]:ob[3], ob[0]:ob[1]]).abs().max().item() / den
            if e != 0.0:
                bad.append((r, i, float("%.1e" % e)))
    q.put((k, alone, bad))


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    ts = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    ctx = mp.get_context("spawn")
    turn, go, q = ctx.Value("i", 0), ctx.Event(), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(k, K, R, L, ts, turn, go, q)) for k in range(K)]
    for p in ps:
        p.start()
    while turn.value < K:
        time.sleep(0.05)
    go.set()
    res = sorted(q.get(timeout=1200) for _ in ps)
    for p in ps:
        p.join(60)
    tag = " ".join(f"{k}={os.environ[k]}" for k in ("MDTILE_REC", "MDTILE_ATTN_MODE", "MDTILE_CONV_MODE", "MDTILE_LIVE_WINDOW", "MDTILE_TILE_BATCH", "MDTILE_FUSE_GN") if k in os.environ)
    nbad = sum(len(b) for _, _, b in res)
    print(f"K={K} R={R} latent {L} tile {ts} [{tag or 'defaults'}]: deterministic alone: {all(a for _, a, _ in res)}; tiles that differ under contention: {nbad} of {K * R * len(res and [0]) or 0}...", flush=True)
    for k, alone, bad in res:
        if bad:
            print(f"   worker {k}: (repeat, tile, rel err) {bad[:12]}")
