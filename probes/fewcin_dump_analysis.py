"""Offline analysis of one failing conv_in call under GPU sharing (round 6).  Input: the dump written by
    FEWCIN_DUMP=gpurun_out/fewcin_dump.npz MDTILE_FEWCIN_FORM=1 python probes/contention_fewcin.py 4 3 mix:handover
(input z, OIHW weights, bias, the indices / values of every output element that differs from the uncontended launch).  For every (wave, cout)
group of wrong elements the script rebuilds the 36 products of the kernel's FMA chain (k = tap * 4 + cin) in float64 and asks which
contiguous run of products, if left out, reproduces the wrong value in all 16 lanes at once.
    python probes/fewcin_dump_analysis.py [dump.npz]        (numpy only, no GPU)"""
import collections, sys
import numpy as np

d = np.load(sys.argv[1] if len(sys.argv) > 1 else "profiles/r6j/fewcin_dump.npz")
z, w, bias, idx, got, ref = d["z"], d["w"], d["bias"], d["idx"], d["got"], d["ref"]
print(f"{len(idx)} wrong elements; x parity {dict(collections.Counter((idx[:, 3] % 2).tolist()))}; lanes {sorted(set(((idx[:, 3] % 128) // 2).tolist()))}")
zp = np.pad(z, ((0, 0), (0, 0), (1, 1), (1, 1))).astype(np.float64)
wk = w.transpose(0, 2, 3, 1).reshape(w.shape[0], 36).astype(np.float64)      # [cout][tap * 4 + cin]
groups = collections.defaultdict(list)
for n, (b, co, r, x) in enumerate(idx):
    groups[(int(b), int(co), int(r), int(x) // 128)].append(n)
print(f"{len(groups)} (batch, cout, row, 128-px block) groups, elements per group: {dict(collections.Counter(len(v) for v in groups.values()))}")
res = collections.Counter()
for key in list(groups)[:600]:
    b, co, r, xb = key
    ns = groups[key]
    V = np.zeros((len(ns), 36))
    for i, n in enumerate(ns):
        x = idx[n, 3]
        for dy in range(3):
            for dx in range(3):
                for ci in range(4):
                    V[i, (dy * 3 + dx) * 4 + ci] = zp[b, ci, r + dy, x + dx]
    terms = V * wk[co][None, :]
    diff = got[ns].astype(np.float64) - ref[ns].astype(np.float64)
    cs = np.concatenate([np.zeros((len(ns), 1)), np.cumsum(terms, 1)], 1)
    best = (1e9, None)
    for j in range(36):
        for k in range(j + 1, 37):
            e = np.abs(diff + (cs[:, k] - cs[:, j])).max()
            if e < best[0]:
                best = (e, (j, k))
    res[f"product {best[1][0]} missing" if best[0] < 2e-5 and best[1][1] == best[1][0] + 1 else (f"products {best[1]} missing" if best[0] < 2e-5 else "unexplained")] += 1
print("what reproduces the wrong value (first 600 groups):")
for k, v in res.most_common():
    print(f"    {v:4d}  {k}")
