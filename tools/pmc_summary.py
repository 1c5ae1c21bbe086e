#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs of the same command) into per-kernel HBM
bytes per launch.   usage: pmc_summary.py <fetch_dir> <write_dir> <out.json>

Units / corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: the counters are in KB; FETCH_SIZE reports
half of the bytes of wide coalesced reads (TCC_EA0_RDREQ x 64 B against 128-B requests) -> doubled; WRITE_SIZE as reported."""
import collections, csv, glob, gzip, json, os, sys


XCDS = 8


MIN_CLOCK_KERNEL_NS = 1.0e6      # GRBM_GUI_ACTIVE counts for the whole counter window around a dispatch: for kernels shorter than ~1 ms the quotient
#                                   comes out at 3 - 17 "GHz" (round 3 printed 3.04 for the 20 us blend); only long kernels fill the window


def load_clock(d):
    """GRBM_GUI_ACTIVE (summed over the 8 XCDs) / kernel duration of the same dispatch -> GHz per kernel, when the pass collected it.
    Kernels whose average dispatch is shorter than MIN_CLOCK_KERNEL_NS, or whose quotient exceeds the chip's 2.4 GHz, get no clock."""
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv*"), recursive=True):
        op = gzip.open if fn.endswith(".gz") else open
        with op(fn, "rt") as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or not r.get("End_Timestamp"):
                    continue
                a = agg[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"]) / XCDS
                a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                a[2] += 1
    return {k: v[0] / v[1] for k, v in agg.items() if v[1] > 0 and v[1] / v[2] >= MIN_CLOCK_KERNEL_NS and v[0] / v[1] <= 2.45}


def libmdtile_digest():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stamp = os.path.join(here, "multidiffusion-upscaler-for-automatic1111_amd", "mdtile", ".libmdtile.stamp")
    return open(stamp).read().strip() if os.path.exists(stamp) else ""


def load(d, counter):
    agg = collections.defaultdict(lambda: [0.0, set()])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv*"), recursive=True)
    for fn in files:
        op = gzip.open if fn.endswith(".gz") else open
        with op(fn, "rt") as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter:
                    continue
                a = agg[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1].add(r["Dispatch_Id"])
    return agg


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    clk = load_clock(fetch_dir)
    kernels = {}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0])[0] * 2 + wr.get(k, [0])[0])):
        nf = len(fe[k][1]) if k in fe else 0
        nw = len(wr[k][1]) if k in wr else 0
        n = max(nf, nw, 1)
        f_kb = fe[k][0] / max(nf, 1) if k in fe else 0.0
        w_kb = wr[k][0] / max(nw, 1) if k in wr else 0.0
        kernels[k[:160]] = {"dispatches": n, "fetch_KB_raw_per_launch": round(f_kb, 1), "write_KB_per_launch": round(w_kb, 1),
                            "hbm_bytes_per_launch": int((2.0 * f_kb + w_kb) * 1024)}
        if k in clk:
            kernels[k[:160]]["clock_GHz"] = round(clk[k], 4)
    with open(out, "w") as f:
        json.dump({"note": "HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB, separate rocprofv3 --pmc passes; clock_GHz = GRBM_GUI_ACTIVE / 8 XCDs / "
                           "kernel duration in the FETCH_SIZE pass, only for kernels of >= 1 ms per dispatch (the counter window of a shorter kernel is mostly not the kernel)", "libmdtile_digest": libmdtile_digest(), "kernels": kernels}, f, indent=1)
    for k, v in list(kernels.items())[:12]:
        print(f"{k[:80]:80s} n={v['dispatches']:5d}  {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
    main()
