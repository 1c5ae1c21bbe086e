#!/usr/bin/env python3
"""Build-time guard for the hand-counted LDS-DMA protocol of the record conv / attention / streaming 1x1 kernels (csrc/vae_conv_rec.hip:
dma16, csrc/vae_attn_bf16x3.hip: dma16a, csrc/vae_conv1x1_bf16x3.hip: dma16s).  The kernels issue `global_load_lds_dwordx4` from inline asm (hipcc would book the builtin as a
FLAT access and degrade every later lgkmcnt wait) and count its completion by hand: M0 is saved, loaded with the LDS destination,
`s_nop 2`, the DMA, M0 restored; every consumer sits behind `s_waitcnt vmcnt(0)` (or the one counted `vmcnt(5)`) + `s_barrier`.
A compiler upgrade that re-orders or drops any of that would only show up as wrong pixels; this script compiles the two files to
device assembly and checks the emitted instruction stream instead:
    * every global_load_lds_dwordx4 is the 4th instruction of   s_mov_b32 sK, m0 / s_mov_b32 m0, sL / s_nop 2 / DMA / s_mov_b32 m0, sK
    * no `s_waitcnt` with a vmcnt between a DMA and its M0 restore, no other M0 writer inside the quintuple
    * every s_barrier is directly preceded (ignoring scalar ALU) by an s_waitcnt, and the vmcnt values those waits count are exactly
      the ones the source asks for: {0, 5} in k_conv3x3_rec (the 5 input pieces of the next K-step may stay in flight at dy = 1),
      {0} in k_upconv_rec and k_attn_bf16x3 (lgkmcnt-only waits -- LDS hand-overs that consume no DMA -- are reported separately),
      {0, 2, 3, 4} in k_conv3x3_rec2 and {0, 6, 7, 8, 9} in k_upconv_rec2 (one counted wait per step position, csrc/vae_conv_rec2.hip),
      {0, 2, 5, 15} in k_conv3x3_recd (probes/csrc/vae_conv_recd.hip, checked with --probes only: the dripped epilogue's slot traffic may stay in flight behind the chunk)
    * MFMA counts per unrolled trip match the source (conv: 36 half-steps x 12; upconv: 24 combo-steps x 12; attention: 24 / slab)
usage: python tools/asm_guard.py   (exit code 0 = ok; prints one line per kernel)      -- also run by tests/test_host_abi.py
"""
from __future__ import annotations

import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd")
CSRC = os.path.join(EXT, "csrc")
# the SAME compiler and code-generation flags as the build (mdtile/build.py), so that the stream checked here is the stream linked
# into libmdtile.so; only the output kind differs (-S, device side only)
sys.path.insert(0, EXT)
from mdtile.build import HIPCC_FLAGS, hipcc_path  # noqa: E402

FLAGS = [f for f in HIPCC_FLAGS if f not in ("-shared",)] + ["-S", "--cuda-device-only"]


def device_asm(src: str, extra=()) -> list:
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run([hipcc] + FLAGS + list(extra) + [src, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed on {src}:\n{r.stdout}")
        return open(out).read().splitlines()


def kernels(lines: list) -> dict:
    """mangled name -> list of instructions (labels, directives and comments stripped)"""
    out, cur = {}, None
    for l in lines:
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", l)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        out[cur].append(t)
        if t == "s_endpgm":
            cur = None
    return out


SREG = r"(?:s\d+|vcc_lo|vcc_hi|ttmp\d+)"


def check_kernel(name: str, ins: list, expect: dict) -> list:
    errs = []
    dma = [k for k, l in enumerate(ins) if l.startswith("global_load_lds_dwordx4")]
    for k in dma:
        if k < 3 or k + 1 >= len(ins):
            errs.append(f"DMA at {k}: no room for the M0 protocol")
            continue
        a, b, c, e = ins[k - 3], ins[k - 2], ins[k - 1], ins[k + 1]
        ma = re.match(r"s_mov_b32 (" + SREG + r"), m0$", a)
        ok = bool(ma) and re.match(r"s_mov_b32 m0, " + SREG + r"$", b) and c.startswith("s_nop") and ma and e == f"s_mov_b32 m0, {ma.group(1)}"
        if not ok:
            errs.append(f"DMA at {k}: expected save-M0 / set-M0 / s_nop / DMA / restore-M0, got {ins[k - 3:k + 2]}")
    bars = [k for k, l in enumerate(ins) if l.startswith("s_barrier")]
    bar_waits = []
    for k in bars:
        # walk back over scalar ALU and branches: a wait selected by a (wave-uniform) branch sits in its own basic block in front of
        # the barrier's block, e.g.  s_waitcnt vmcnt(0) / loop header / s_cbranch / s_waitcnt vmcnt(5) / s_barrier
        j, found = k - 1, []
        # (v_readlane / v_writelane: SGPR spill traffic of the scalar address arithmetic -- register moves, no memory; plain integer VALU
        # address arithmetic that hipcc schedules between the wait and the barrier is harmless as well: it touches no memory and no counter)
        while j >= 0 and (re.match(r"s_(mov|movk|add|and|andn2|or|lshl|lshr|mul|cmp|cselect|sub|ashr|bfe|nop|xor|not|max|min|addc|bitcmp|cbranch|branch|waitcnt|load_dword)"
                                   r"|v_readlane_b32|v_writelane_b32|v_(mov|and|or|lshlrev|lshrrev|add|sub|and_or|lshl_add|lshl_or|mad_u32_u24|mul_u32_u24|bfe_u32|add_lshl)_", ins[j])):
            if ins[j].startswith("s_waitcnt"):
                found.append(ins[j])
            j -= 1
        if not found:
            errs.append(f"s_barrier at {k} is not preceded by an s_waitcnt (found {ins[max(0, j)]})")
            continue
        for w in found:
            m = re.search(r"vmcnt\((\d+)\)", w)
            bar_waits.append(int(m.group(1)) if m else None)      # None: an lgkmcnt-only wait (LDS hand-over, no DMA consumed behind it)
    counted = {w for w in bar_waits if w is not None}
    n_mfma = sum(1 for l in ins if l.startswith("v_mfma_f32_32x32x16_bf16"))
    for key, val in expect.items():
        if key == "dma_min" and len(dma) < val:
            errs.append(f"{len(dma)} DMA instructions < {val}")
        if key == "barrier_vmcnt" and counted != set(val):
            errs.append(f"the waits in front of the barriers count vmcnt {sorted(counted)}, the DMA protocol expects exactly {sorted(val)}")
        if key == "mfma_multiple" and (n_mfma == 0 or n_mfma % val):
            errs.append(f"{n_mfma} MFMAs is not a multiple of {val}")
    print(f"{name[:70]:70s} {len(ins):6d} instr  {len(dma):3d} DMA  {len(bars):3d} barriers (vmcnt in front: {sorted(counted)}, lgkm-only: {bar_waits.count(None)})  "
          f"{n_mfma:4d} MFMA  {'ok' if not errs else 'FAIL'}")
    return [f"{name}: {e}" for e in errs]


def main() -> int:
    errs = []
    rec = kernels(device_asm(os.path.join(CSRC, "vae_conv_rec.hip")))
    rec2 = kernels(device_asm(os.path.join(CSRC, "vae_conv_rec2.hip")))
    # --probes: also the kernels that exist in the PROBES twin only (probes/csrc/: the dripped-epilogue conv, a measured and rejected form)
    probes_csrc = os.path.join(ROOT, "probes", "csrc")
    recd = kernels(device_asm(os.path.join(probes_csrc, "vae_conv_recd.hip"), ["-I" + CSRC])) if "--probes" in sys.argv else None
    att = kernels(device_asm(os.path.join(CSRC, "vae_attn_bf16x3.hip")))
    c11 = kernels(device_asm(os.path.join(CSRC, "vae_conv1x1_bf16x3.hip")))
    plan = [
        (c11, "k_conv1x1_streamILi2", dict(dma_min=5, barrier_vmcnt=[0, 5], mfma_multiple=24)),
        (c11, "k_conv1x1_streamILi4", dict(dma_min=4, barrier_vmcnt=[0, 4], mfma_multiple=24)),
        (rec, "k_conv3x3_recILi2ELi2ELi4", dict(dma_min=8, barrier_vmcnt=[0, 5], mfma_multiple=12)),
        (rec, "k_conv3x3_recILi1ELi1ELi2", dict(dma_min=4, barrier_vmcnt=[0, 5], mfma_multiple=3)),
        (rec, "k_upconv_recE", dict(dma_min=8, barrier_vmcnt=[0], mfma_multiple=12)),
        # the same two kernel texts + statistics in the epilogue (slow mode's pooled sites): the protocol is the shipping kernels'
        (rec, "k_conv3x3_rec_stILi2ELi2ELi4", dict(dma_min=8, barrier_vmcnt=[0, 5], mfma_multiple=12)),
        (rec, "k_upconv_rec_stE", dict(dma_min=8, barrier_vmcnt=[0], mfma_multiple=12)),
        # two blocks per CU: one counted wait per step position (tools/rec2_protocol_sim.py derives and checks the values)
        (rec2, "k_conv3x3_rec2ILi2ELi2ELi4", dict(dma_min=8, barrier_vmcnt=[0, 2, 3, 4], mfma_multiple=12)),
        (rec2, "k_upconv_rec2E", dict(dma_min=8, barrier_vmcnt=[0, 6, 7, 8, 9], mfma_multiple=12)),
        (att, "k_attn_bf16x3ILi512", dict(dma_min=8, barrier_vmcnt=[0], mfma_multiple=6)),
        (att, "k_attn_bf16x3ILi256", dict(dma_min=8, barrier_vmcnt=[0], mfma_multiple=6)),
        (att, "k_attn_bf16x3ILi128", dict(dma_min=8, barrier_vmcnt=[0], mfma_multiple=6)),
    ]
    if recd is not None:      # dripped epilogue: vmcnt(5) at dy = 1, and behind a slot phase a lower bound of the slot's own memory instructions (15 / 2)
        plan.append((recd, "k_conv3x3_recdE", dict(dma_min=8, barrier_vmcnt=[0, 2, 5, 15], mfma_multiple=12)))
    for table, sub, expect in plan:
        hit = [n for n in table if sub in n]
        if not hit:
            errs.append(f"kernel {sub} not found in the device assembly")
            continue
        errs += check_kernel(hit[0], table[hit[0]], expect)
    # conv_in (csrc/vae_conv.hip: k_conv3x3_fewcin): no packed-fp32 instruction of the shipping form may read the HIGH register of a pair for
    # the LOW half of its result (`op_sel:[..1..]`) -- the encoding that dropped single products when the kernel shared a CU with another
    # kernel's MFMA waves (round 6, profiles/r6j).  The shipping form keeps every weight in LDS as a ready-made pair and needs none.
    cin = kernels(device_asm(os.path.join(CSRC, "vae_conv.hip")))
    for cinv in (3, 4):
        hit = [n for n in cin if f"k_conv3x3_fewcinILi{cinv}ELi0E" in n]
        if not hit:
            errs.append(f"kernel k_conv3x3_fewcin<{cinv}, 0> not found in the device assembly")
            continue
        ins = cin[hit[0]]
        pk = [l for l in ins if l.startswith("v_pk_")]
        bad = [l for l in pk if re.search(r"op_sel:\[", l)]
        print(f"{hit[0][:70]:70s} {len(ins):6d} instr  {len(pk):3d} packed-fp32 instructions, {len(bad)} with op_sel  {'ok' if pk and not bad else 'FAIL'}")
        if not pk or bad:
            errs.append(f"{hit[0]}: {len(bad)} packed instructions carry op_sel (first: {bad[:1]})")
    for e in errs:
        print("ASM GUARD:", e)
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
