#!/usr/bin/env python3
"""Per-kernel averages of SQ / GRBM counters from one or more rocprofv3 --pmc passes.   usage: pmc_sq.py <out.json> <dir> [<dir> ...]
SQ_* are quad-cycles summed over waves except SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over SIMDs): for a 32x32x16 bf16 MFMA it
is 32 x the MFMAs issued.  Derived: MFMA-pipe busy fraction = BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel time x clock), with the
clock taken both at the nominal 2.4 GHz and from GRBM_GUI_ACTIVE / kernel time when that counter was collected (rocprofv3 reports
GRBM_GUI_ACTIVE summed over the 8 XCDs: divided by XCDS here)."""
import collections, csv, glob, gzip, json, os, sys

XCDS = 8


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    dur = collections.defaultdict(lambda: [0.0, 0])
    for d in dirs:
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv*"), recursive=True):
            op = gzip.open if fn.endswith(".gz") else open
            with op(fn, "rt") as f:
                for r in csv.DictReader(f):
                    a = agg[r["Kernel_Name"]][r["Counter_Name"]]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
                    if "Start_Timestamp" in r and r.get("End_Timestamp"):
                        t = dur[(r["Kernel_Name"], r["Counter_Name"])]
                        t[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                        t[1] += 1
    res = {}
    for k, cs in agg.items():
        e = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        e["dispatches"] = max(v[1] for v in cs.values())
        ns = [dur[(k, c)][0] / max(dur[(k, c)][1], 1) for c in cs if dur[(k, c)][1]]
        if ns:
            e["avg_ns"] = sum(ns) / len(ns)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                e["mfma_busy_frac_at_2.4GHz"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * e["avg_ns"] * 2.4)
            if "GRBM_GUI_ACTIVE" in e:
                e["clock_GHz"] = e["GRBM_GUI_ACTIVE"] / XCDS / e["avg_ns"]
                if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                    e["mfma_busy_frac_at_clock"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * e["GRBM_GUI_ACTIVE"] / XCDS)
        if "SQ_WAVE_CYCLES" in e:
            for nm, c in (("wave_parked_frac", "SQ_WAIT_ANY"), ("wave_issue_stall_frac", "SQ_WAIT_INST_ANY"), ("wave_issuing_frac", "SQ_ACTIVE_INST_ANY")):
                if c in e:
                    e[nm] = e[c] / e["SQ_WAVE_CYCLES"]
        res[k[:150]] = e
    digest = None
    try:      # source digest of the library the counted process loaded (the same tree: gpurun snapshots it), as tools/pmc_summary.py records it
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multidiffusion-upscaler-for-automatic1111_amd"))
        from mdtile import build
        digest = build._digest()[:12]
    except Exception:
        pass
    with open(out, "w") as f:
        json.dump({"libmdtile_digest": digest, "kernels": res}, f, indent=1)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("avg_ns", 0))[:8]:
        print(k[:70], {a: (round(b, 3) if isinstance(b, float) and b < 100 else b) for a, b in e.items() if a in (
            "dispatches", "avg_ns", "mfma_busy_frac_at_2.4GHz", "clock_GHz", "mfma_busy_frac_at_clock", "wave_parked_frac", "wave_issue_stall_frac", "wave_issuing_frac")})


if __name__ == "__main__":
    main()
