"""Which ENGINE CALL gives different results when other processes load the same GPU?  (Follow-up of probes/contention_determinism.py.)
K processes share cuda:0; each first computes every op's reference alone (one process at a time), then all loop over the ops R times and
compare bit for bit.   python probes/contention_ops.py [K] [R]"""
import os, sys, time
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_ops(E, dev):
    torch.manual_seed(0)
    ops = {}

    def conv(cin, cout, k=3):
        c = torch.nn.Conv2d(cin, cout, k, padding=k // 2)
        return E.PackedConv(c.weight.detach().to(dev), c.bias.detach().to(dev))

    def coef(B, C, seed):
        g = torch.Generator().manual_seed(seed)
        return torch.cat([torch.rand(B, 1, C, generator=g) * 1.5 + 0.25, torch.randn(B, 1, C, generator=g) * 0.5], dim=1).contiguous().to(dev)

    # the decoder's shapes at one 278^2 tile (reduced planes where the full one is only more of the same)
    x512 = torch.randn(2, 512, 278, 278, device=dev)
    x256 = torch.randn(1, 256, 1112, 1112, device=dev)
    x128 = torch.randn(1, 128, 1200, 1056, device=dev)
    z4 = torch.randn(2, 4, 278, 278, device=dev)
    pc_in = conv(4, 512); pc512 = conv(512, 512); pc256 = conv(256, 256); pc128 = conv(128, 128); pc_out = conv(128, 3)
    pc1_512 = conv(512, 512, 1); pc1_nin = conv(512, 256, 1); pc1_nin2 = conv(256, 128, 1)
    c512, c256, c128 = coef(2, 512, 1), coef(1, 256, 2), coef(1, 128, 3)
    r512 = torch.randn(2, 512, 278, 278, device=dev)
    ops["conv_in fewcin 4->512"] = lambda: pc_in(z4)
    ops["handover gn conv 512->512 278^2"] = lambda: pc512(x512, residual=r512, pre_gn=c512)
    ops["handover gn conv 256->256 1112^2"] = lambda: pc256(x256, pre_gn=c256)
    ops["handover gn+stats conv 128->128"] = lambda: torch.cat([t.flatten().float() for t in (lambda y, s: (y, s[0], s[1]))(*pc128.call_stats(x128, c128))])
    xr512 = E.rec_from_f32(x512, c512); xr256 = E.rec_from_f32(x256, c256); xr128 = E.rec_from_f32(x128, c128)
    ops["rec_from_f32 512 278^2"] = lambda: E.rec_from_f32(x512, c512).records()
    ops["rec conv 512->512 278^2 (+res, f32+rec)"] = lambda: (lambda y, yr: torch.cat([y.flatten().view(torch.int32), yr.records().flatten()]))(*pc512.call_rec(xr512, residual=r512, want_f32=True, want_rec=True, rec_coef=c512))
    ops["rec conv 256->256 1112^2 (rec only)"] = lambda: pc256.call_rec(xr256, want_f32=False, want_rec=True, rec_coef=c256)[1].records()
    ops["rec conv 128->128 1200x1056 (f32)"] = lambda: pc128.call_rec(xr128, want_f32=True)[0]
    ops["rec conv_out 128->3"] = lambda: pc_out.call_rec(xr128, want_f32=True)[0]
    ops["rec upconv 512->512 278^2 -> 556^2"] = lambda: (lambda y, yr: torch.cat([y.flatten().view(torch.int32), yr.records().flatten()]))(*pc512.call_rec(E.rec_from_f32(x512, None), upsample2x=True, want_f32=True, want_rec=True))
    ops["rec upconv window"] = lambda: pc512.call_rec(E.rec_from_f32(x512, None), upsample2x=True, want_f32=True, want_rec=False, window=([3, 5], [2, 7], 260, 262))[0]
    ops["rec conv + stats 512->512"] = lambda: torch.cat([t.flatten().float() for t in (lambda y, s: (y, s[0], s[1]))(*pc512.call_rec_stats(xr512, residual=r512))])
    ops["conv1x1 stream<4> 512->512 278^2 (+res)"] = lambda: pc1_512(x512, residual=r512)
    ops["conv1x1 stream<4> 512->256 1112^2"] = lambda: pc1_nin(torch.randn(1, 512, 1112, 1112, generator=None, device=dev) * 0 + x256.repeat(1, 2, 1, 1))
    ops["conv1x1 stream<2> 256->128 1112^2"] = lambda: pc1_nin2(x256)
    q, k, v = (torch.randn(1, 512, 20000, device=dev) for _ in range(3))
    ops["attention bf16x3 T=20000"] = lambda: E.vae_attn(q, k, v, 512 ** -0.5, v_channel_major=True)
    ops["gn_stats 512"] = lambda: torch.cat(E.gn_stats(x512, 32))
    v_, m_ = E.gn_stats(x512, 32)
    g = torch.rand(512, device=dev) + 0.5; bt = torch.randn(512, device=dev)
    ops["gn_apply 512"] = lambda: E.gn_apply(x512, m_, v_, g, bt, 32, 1e-6, False)
    ops["gn_coeffs"] = lambda: E.gn_coeffs(m_, v_, g, bt, 512, 32, 1e-6)
    zbig = torch.randn(1, 4, 1024, 1024, device=dev)
    ops["vae_fast_input"] = lambda: E.vae_fast_input(zbig, 256)
    return ops


def worker(k, K, R, turn, go, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))
    import mdtile as E
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    ops = make_ops(E, dev)
    while turn.value != k:
        time.sleep(0.01)
    refs, alone_bad = {}, []
    for name, fn in ops.items():
        a = fn().clone(); b = fn()
        if not torch.equal(a, b):
            alone_bad.append(name)
        refs[name] = a
    torch.cuda.synchronize()
    with turn.get_lock():
        turn.value += 1
    go.wait()
    bad, detail = {}, {}
    for r in range(R):
        for name, fn in ops.items():
            if name.startswith("conv_in"):      # were the differing elements WRITTEN at all?  hand the allocator a NaN-filled block of the output's size first
                t = torch.full((2, 512, 278, 278), float("nan"), device="cuda:0")
                del t
            out = fn()
            if not torch.equal(out, refs[name]):
                bad[name] = bad.get(name, 0) + 1
                if name not in detail:
                    ne = (out != refs[name])
                    idx = ne.nonzero()
                    d = (out.double() - refs[name].double()).abs()
                    nan_note = f" [{int(torch.isnan(out.float()).sum())} NaN = never written]" if out.is_floating_point() else ""
                    detail[name] = (nan_note + f"shape {tuple(out.shape)}: {int(ne.sum())} of {out.numel()} elements differ, max |d| {d.max().item():.3g} (ref max {refs[name].double().abs().max().item():.3g}); "
                                    f"index min {idx.min(0).values.tolist()} max {idx.max(0).values.tolist()}; first {idx[0].tolist()}")
    q.put((k, alone_bad, bad, detail))


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    ctx = mp.get_context("spawn")
    turn, go, q = ctx.Value("i", 0), ctx.Event(), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(k, K, R, turn, go, q)) for k in range(K)]
    for p in ps:
        p.start()
    while turn.value < K:
        time.sleep(0.05)
    go.set()
    res = sorted(q.get(timeout=1800) for _ in ps)
    for p in ps:
        p.join(60)
    tot = {}
    for k, alone_bad, bad, detail in res:
        for n, dsc in detail.items():
            print(f"worker {k}: {n}: {dsc}")
        if alone_bad:
            print(f"worker {k}: NOT deterministic even alone: {alone_bad}")
        for n, c in bad.items():
            tot[n] = tot.get(n, 0) + c
    print(f"K={K} processes x R={R} repeats: calls whose result differed from the uncontended one (count of {K * R}):")
    for n, c in sorted(tot.items(), key=lambda t: -t[1]):
        print(f"   {c:3d}  {n}")
    if not tot:
        print("   none")
