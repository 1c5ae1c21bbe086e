"""The probe scripts drive switches that exist only in the PROBES twin of the library (csrc/common.h: kProbes; MDTILE_REC_DBG, _STAMPS,
_GRID, _BLOCKS, _PERSIST, _STAGGER_PCT, MDTILE_REC2_*, MDTILE_BLEND_CFG, MDTILE_ATTN_SPLIT, MDTILE_C1X1_STREAM).  `use(E)` points the
ctypes binding at probes/_ab/libmdtile_probes.so (built here with hipcc when missing: python -m mdtile.build --probes) BEFORE the first
call loads a library.  The shipping libmdtile.so reads none of these."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd"))


CONV_REC_DRIP = 16      # mdtile_conv2d_rec flag bit of the dripped-epilogue kernel (probes/csrc/vae_conv_recd.hip): the PROBES twin only


def use(E):
    from mdtile import build as b
    E.CONV_REC_DRIP = CONV_REC_DRIP
    if not os.path.exists(b.PROBES_LIB) or os.path.getmtime(b.PROBES_LIB) < max(os.path.getmtime(f) for f in b._sources() + b._probe_sources()):
        b.build_probes()
    assert E._lib is None, "probes: the library is already loaded"
    E.LIB_PATH = b.PROBES_LIB
    print(f"[probes] using {b.PROBES_LIB}", file=sys.stderr)
