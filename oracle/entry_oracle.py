"""
ORACLE (test infrastructure only) -- CPU restatement of MultiDiffusion's ENTRY path: what upstream runs between the sampler's call
of `model_wrap_cfg.inner_model.forward(x, sigma, cond=...)` and the blend, i.e. `hook` -> `kdiff_forward` / `ddim_forward` ->
`sample_one_step` with `repeat_func` / `custom_func` -> `repeat_tensor` / `repeat_cond_dict` (tile_methods/multidiffusion.py:15-29,
52-129) and the whole-batch branch of the per-region forwards (tile_methods/abstractdiffusion.py:231-287, 429-451).

Only tests/ may import this file.  Parity status: PINNED -- tests/test_md_entry_path.py runs the upstream delegate itself (hook() and
three sampler steps on a conditioning-dependent stand-in model) under hostsim/stub_host.py next to these functions with torch.equal,
and tests/golden/entry.npz (made by tests/golden/make_golden_entry.py from the upstream code) carries the pin to the GPU box.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch

from oracle import blend_oracle as bo


# ---- cond dict access (abstractdiffusion.py:120-166) ----------------------------------------------------------------------
def _tkey(cond) -> str:
    return "crossattn" if "crossattn" in cond else "c_crossattn"


def get_tcond(cond):
    t = cond[_tkey(cond)]
    return t[0] if isinstance(t, list) else t


def get_icond(cond):
    i = cond["c_concat"]          # stub host: conditioning_key == "crossattn" -> 'c_concat' (:133-134)
    return i[0] if isinstance(i, list) else i


def make_cond_dict(cond_in, tcond, icond, vcond=None):
    """:160-166 -- copy, then replace text / image / vector conditioning (list-wrapped where the input was)."""
    out = cond_in.copy()
    k = _tkey(out)
    out[k] = [tcond] if isinstance(out[k], list) else tcond
    out["c_concat"] = [icond] if isinstance(out["c_concat"], list) else icond
    if "vector" in out:
        out["vector"] = vcond
    return out


# ---- batching helpers (multidiffusion.py:100-129) -----------------------------------------------------------------------------
def repeat_tensor(x: torch.Tensor, n: int) -> torch.Tensor:
    """:100-110 -- expand a batch-1 tensor, tile a larger batch, along dim 0."""
    if n == 1:
        return x
    r = len(x.shape) - 1
    if x.shape[0] == 1:
        return x.expand([n] + [-1] * r)
    return x.repeat([n] + [1] * r)


def repeat_cond_dict(cond_in: Dict, batch: Sequence, H: int, W: int) -> Dict:
    """:112-129 -- text cond repeated per tile; image cond SLICED per bbox when it has the latent's spatial size (img2img), else
    repeated; SDXL vector cond repeated."""
    n = len(batch)
    tcond = repeat_tensor(get_tcond(cond_in), n)
    icond = get_icond(cond_in)
    if tuple(icond.shape[2:]) == (H, W):
        icond = torch.cat([icond[:, :, y:y + th, x:x + tw] for (x, y, tw, th) in batch], dim=0)
    else:
        icond = repeat_tensor(icond, n)
    vcond = cond_in.get("vector")
    if vcond is not None:
        vcond = repeat_tensor(vcond, n)
    return make_cond_dict(cond_in, tcond, icond, vcond)


# ---- the hijacked forwards --------------------------------------------------------------------------------------------------
def kdiff_forward(o: bo.BlendOracle, x_in, sigma_in, cond, forward: Callable, step: int = 0,
                  region_conds: Optional[List] = None) -> torch.Tensor:
    """multidiffusion.py:52-73 (`kdiff_forward`) around `sample_one_step` (:131-218, the arithmetic is BlendOracle.evaluate).
    region_conds[i] = (cond tensor fn(step), uncond tensor fn(step)) of region i: the whole-batch / equal-token-length branch of
    kdiff_custom_forward (abstractdiffusion.py:262-287: cond = cat([tensor, uncond]); image cond cut to the region)."""
    if tuple(x_in.shape[2:]) != (o.H, o.W):                       # :140-144 -- hires pass: untiled
        return forward(x_in, sigma_in, cond=cond)

    def tile_fn(x_tile, batch):                                   # repeat_func, :60-66
        return forward(x_tile, repeat_tensor(sigma_in, len(batch)), cond=repeat_cond_dict(cond, batch, o.H, o.W))

    def region_fn(x_r, i, r):                                     # custom_func, :68-69
        tensor, uncond = region_conds[i][0](step), region_conds[i][1](step)
        icond = get_icond(cond)
        if tuple(icond.shape[2:]) == (o.H, o.W):
            icond = icond[r.sl]
        return forward(x_r, sigma_in, cond=make_cond_dict(cond, torch.cat([tensor, uncond]), icond))

    return o.evaluate(x_in, tile_fn, region_fn, with_boxes=True)


def ddim_forward(o: bo.BlendOracle, x_in, ts_in, cond, forward: Callable, step: int = 0, region_conds: Optional[List] = None) -> torch.Tensor:
    """multidiffusion.py:75-98 (`ddim_forward`): cond is a dict (repeat_cond_dict) or a bare tensor (repeat_tensor); a region goes
    through ddim_custom_forward (abstractdiffusion.py:429-451): uncond padded with its last token / truncated to the prompt's length,
    both wrapped back into cond dicts, forward(x, cond, ts, unconditional_conditioning=uncond)."""
    if tuple(x_in.shape[2:]) != (o.H, o.W):
        return forward(x_in, ts_in, cond=cond)

    def tile_fn(x_tile, batch):
        n = len(batch)
        cond_tile = repeat_cond_dict(cond, batch, o.H, o.W) if isinstance(cond, dict) else repeat_tensor(cond, n)
        return forward(x_tile, repeat_tensor(ts_in, n), cond=cond_tile)

    def region_fn(x_r, i, r):
        tensor, uncond = region_conds[i][0](step), region_conds[i][1](step)
        icond = None
        if isinstance(cond, dict):
            icond = get_icond(cond)
            if tuple(icond.shape[2:]) == (o.H, o.W):
                icond = icond[r.sl]
        if uncond.shape[1] < tensor.shape[1]:
            uncond = torch.hstack([uncond, uncond[:, -1:].repeat([1, tensor.shape[1] - uncond.shape[1], 1])])
        elif uncond.shape[1] > tensor.shape[1]:
            uncond = uncond[:, :tensor.shape[1]]
        c, uc = tensor, uncond
        if icond is not None:
            c, uc = make_cond_dict(cond, tensor, icond), make_cond_dict(cond, uncond, icond)
        return forward(x_r, c, ts_in, unconditional_conditioning=uc)

    return o.evaluate(x_in, tile_fn, region_fn, with_boxes=True)
