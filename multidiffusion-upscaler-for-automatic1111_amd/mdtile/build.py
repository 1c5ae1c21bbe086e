"""Builds libmdtile.so (all HIP kernels + the C ABI) in-tree with hipcc for gfx950.

    python -m mdtile.build          (from the extension root)   or   __graft_entry__.build()

The .so is git-ignored but travels with the working tree (A1111 extension dir / gpurun snapshot).
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
EXT_ROOT = os.path.dirname(HERE)
CSRC = os.path.join(EXT_ROOT, "csrc")
INCLUDE = os.path.join(os.path.dirname(EXT_ROOT), "include")
LIB = os.path.join(HERE, "libmdtile.so")
STAMP = os.path.join(HERE, ".libmdtile.stamp")

# -ffp-contract=off: the blend reproduces eager-torch op-by-op fp32 rounding (see csrc/blend.hip); kernels that
# want FMAs ask for them explicitly (fmaf / MFMA builtins).
# -pragma-unroll-threshold: the record conv kernels (csrc/vae_conv_rec.hip) keep two fragment register sets whose indices are
# compile-time only after their 36- / 24-step K-loop bodies are FULLY unrolled; the default budget (16 K instructions, estimated
# before constant folding) is too small for that and a partial unroll would turn the register arrays into scratch memory.
# -DMDTILE_PROBES=0: the shipping library carries no probe scaffolding (csrc/common.h: kProbes / probe_env); build_probes() below
# makes the instrumented twin the scripts under probes/ load.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-mllvm", "-pragma-unroll-threshold=262144", "-DMDTILE_PROBES=0"]
PROBES_LIB = os.path.join(os.path.dirname(EXT_ROOT), "probes", "_ab", "libmdtile_probes.so")


PROBES_CSRC = os.path.join(os.path.dirname(EXT_ROOT), "probes", "csrc")


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _probe_sources():
    """Kernels that exist in the PROBES twin only (measured and rejected forms kept for their A/B scripts): probes/csrc/*.hip."""
    return sorted(glob.glob(os.path.join(PROBES_CSRC, "*.hip")))


def _digest() -> str:
    h = hashlib.sha256()
    for f in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(INCLUDE, "mdtile.h")]:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode())   # not the absolute path: the tree is copied around (gpurun snapshot, extensions/)
            h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def hipcc_path() -> str | None:
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile if sources changed; returns the path of the shared library."""
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == digest:
        return LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: libmdtile.so cannot be built (ROCm toolchain required)")
    if os.path.exists(STAMP):
        os.remove(STAMP)      # a build that fails half way (link done, assembly guard rejected) must not leave the OLD digest next to a NEW library
    objs = []
    objdir = os.path.join(EXT_ROOT, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc] + [f for f in HIPCC_FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    # build-time guard of the hand-counted LDS-DMA protocol (M0 save / set / restore around every global_load_lds, the counted vmcnt
    # waits in front of the barriers): checked on the emitted device assembly, so a compiler upgrade cannot break it silently
    guard = os.path.join(os.path.dirname(EXT_ROOT), "tools", "asm_guard.py")
    if os.path.exists(guard) and os.environ.get("MDTILE_SKIP_ASM_GUARD", "") != "1":
        g = subprocess.run([sys.executable, guard], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if g.returncode != 0:
            raise RuntimeError("tools/asm_guard.py rejected the device code of the record conv / attention kernels (it checks hipcc's assembly for the "
                               "hand-counted LDS-DMA protocol; after a compiler upgrade a mismatch of its patterns looks the same as a real violation -- "
                               f"MDTILE_SKIP_ASM_GUARD=1 builds without the check):\n{g.stdout}")
        if verbose:
            print(g.stdout.rstrip())
    with open(STAMP, "w") as f:
        f.write(digest)
    if verbose:
        print(f"[mdtile] built {LIB} from {len(objs)} sources")
    return LIB


def build_probes(verbose: bool = True) -> str:
    """The PROBES twin of the library (the shipping sources + probes/csrc/*.hip, -DMDTILE_PROBES=1: MDTILE_REC_DBG / _STAMPS / _GRID / _BLOCKS / _PERSIST / _STAGGER_PCT,
    MDTILE_REC2_*, MDTILE_BLEND_CFG, MDTILE_ATTN_SPLIT, MDTILE_C1X1_STREAM are read per launch) -> probes/_ab/libmdtile_probes.so
    (git-ignored, travels with gpurun).  probes/_probes_lib.py: use(E) points the ctypes binding at it before the first call; never shipped, never the default."""
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found")
    os.makedirs(os.path.dirname(PROBES_LIB), exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-DMDTILE_PROBES=0"] + ["-DMDTILE_PROBES=1", "-I" + CSRC]
    objdir = os.path.join(EXT_ROOT, "build", "probes")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in _sources() + _probe_sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen([hipcc] + [f for f in flags if f != "-shared"] + ["-c", src, "-o", obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBES_LIB] + objs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    if verbose:
        print(f"[mdtile] built {PROBES_LIB} (probes twin)")
    return PROBES_LIB


if __name__ == "__main__":
    if "--probes" in sys.argv:
        build_probes()
    else:
        build(force="--force" in sys.argv)
