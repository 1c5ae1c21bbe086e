#!/usr/bin/env python3
"""Regenerates tests/golden/entry.npz by running the UPSTREAM MultiDiffusion delegate (hook() + three sampler steps through
`inner_model.forward`, tile_methods/multidiffusion.py:15-29, 52-129) under hostsim/stub_host.py -- see tests/entry_driver.py for the
stand-in model and the cases.      python tests/golden/make_golden_entry.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from hostsim import stub_host as sh  # noqa: E402
import entry_driver as ed  # noqa: E402


def main():
    ref = sh.load_reference()
    out = {}
    for case in ed.ENTRY_CASES:
        outs, calls = ed.drive(ref, case, "cpu", ed.ref_regions(ref))
        out[case["name"] + "/outs"] = outs.numpy()
        out[case["name"] + "/calls"] = np.array([list(c[1:]) for c in calls], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "entry.npz"), **out)
    print("entry.npz", os.path.getsize(os.path.join(HERE, "entry.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
