#!/usr/bin/env python3
"""Did a source change alter the code of a kernel that was meant to stay as it is?   usage: python tools/isa_diff.py <git-rev> [file.hip ...]
Compiles the named csrc/*.hip files (default: the conv / attention kernel files) to gfx950 device assembly twice -- as they are in the working
tree and as they were at <git-rev> (checked out into a scratch directory with the headers of that revision) -- with the build's own flags
(mdtile/build.py: HIPCC_FLAGS), and compares every kernel both sides have instruction by instruction.  Basic-block label numbers are
normalised (they count functions of the file); a template parameter a kernel gained with a default value (`..., false>`) is matched to the
old name.  Prints one line per kernel: IDENTICAL, or the number of differing lines and the first few.
Round 5 used it to show that the statistics variants of the record kernels (k_conv3x3_rec_st, k_upconv_rec_st: the same text included a
second time) left the shipping kernels' code untouched: the only difference is the kernarg offset of the implicit arguments (0xd0 -> 0xd8,
ConvRParams grew by one pointer).  CPU-only (hipcc cross-compiles)."""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "multidiffusion-upscaler-for-automatic1111_amd"
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, PKG))
import asm_guard as ag                      # noqa: E402
from mdtile.build import HIPCC_FLAGS, hipcc_path   # noqa: E402

DEFAULT = ["vae_conv_rec.hip", "vae_conv_rec2.hip", "vae_conv_bf16x3.hip", "vae_conv1x1_bf16x3.hip", "vae_attn_bf16x3.hip", "blend.hip"]


def asm(src: str) -> dict:
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-S", "--cuda-device-only"]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run([hipcc_path()] + flags + [src, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed on {src}:\n{r.stdout}")
        return ag.kernels(open(out).read().splitlines())


def checkout(rev: str, into: str) -> None:
    """csrc/ and include/ of <rev> in the same relative layout (csrc includes ../../include/mdtile.h)"""
    for sub in (f"{PKG}/csrc", "include"):
        os.makedirs(os.path.join(into, sub), exist_ok=True)
        names = subprocess.run(["git", "-C", ROOT, "ls-tree", "--name-only", f"{rev}:{sub}"], capture_output=True, text=True, check=True).stdout.split()
        for n in names:
            blob = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{sub}/{n}"], capture_output=True, check=True).stdout
            with open(os.path.join(into, sub, n), "wb") as f:
                f.write(blob)


def norm(line: str) -> str:
    return re.sub(r"\.LBB\d+_", ".LBBx_", line)


def main() -> int:
    if len(sys.argv) < 2:
        print(__doc__)
        return 2
    rev, files = sys.argv[1], sys.argv[2:] or DEFAULT
    changed = 0
    with tempfile.TemporaryDirectory() as old_root:
        checkout(rev, old_root)
        for f in files:
            old_src = os.path.join(old_root, PKG, "csrc", f)
            if not os.path.exists(old_src):
                print(f"{f}: not in {rev}")
                continue
            old, new = asm(old_src), asm(os.path.join(ROOT, PKG, "csrc", f))
            alias = {re.sub(r"ELb0EEEv", "EEEv", k): k for k in new}      # a new trailing `false` template argument
            for k, a in old.items():
                kk = k if k in new else alias.get(k)
                if kk is None:
                    print(f"{f}: {k[:90]}  -- gone")
                    changed += 1
                    continue
                b = new[kk]
                d = [(x, y) for x, y in zip(map(norm, a), map(norm, b)) if x != y]
                if len(a) == len(b) and not d:
                    print(f"{f}: {k[:90]}  {len(a)} instr  IDENTICAL")
                else:
                    changed += 1
                    print(f"{f}: {k[:90]}  {len(a)} -> {len(b)} instr, {len(d)} differing lines: {d[:3]}")
            for k in new:
                if k not in old and re.sub(r"ELb0EEEv", "EEEv", k) not in old:
                    print(f"{f}: {k[:90]}  -- new ({len(new[k])} instr)")
    return 1 if changed else 0


if __name__ == "__main__":
    sys.exit(main())
