"""The REAL MultiDiffusion entry path: `delegate.hook()` -> `sampler.model_wrap_cfg.inner_model.forward(x, sigma, cond=...)` ->
`kdiff_forward` / `ddim_forward` -> `repeat_func` -> `repeat_tensor` / `repeat_cond_dict` -> blend, three sampler steps with a
stand-in UNet that depends on tile content, sigma, c_crossattn rows, c_concat (latent-sized => sliced per bbox) and SDXL's vector.

  * CPU (`-m "not gpu"`): the upstream delegate itself (tile_methods/multidiffusion.py:15-29, 52-129 under hostsim/stub_host.py) against
    oracle/entry_oracle.py AND the committed tests/golden/entry.npz -- pins the restatement;
  * GPU (`-m gpu`): this repo's delegate (mdtile engine) against the oracle and the upstream-made goldens, torch.equal.
"""
import os

import numpy as np
import pytest
import torch

import entry_driver as ed
from hostsim import stub_host as sh

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "entry.npz")
IDS = [c["name"] for c in ed.ENTRY_CASES]


@pytest.fixture(scope="module")
def golden_entry():
    return np.load(GOLDEN)


@pytest.mark.parametrize("case", ed.ENTRY_CASES, ids=IDS)
def test_oracle_matches_upstream_goldens(case, golden_entry):
    outs, calls = ed.oracle_run(case)
    assert np.array_equal(outs.numpy(), golden_entry[case["name"] + "/outs"])
    assert [list(c[1:]) for c in calls] == golden_entry[case["name"] + "/calls"].tolist()


@pytest.mark.skipif(not sh.reference_available(), reason="/root/reference not mounted")
@pytest.mark.parametrize("case", ed.ENTRY_CASES, ids=IDS)
def test_oracle_matches_upstream_entry_path(case):
    ref = sh.load_reference()
    want, calls_ref = ed.drive(ref, case, "cpu", ed.ref_regions(ref))
    got, calls = ed.oracle_run(case)
    assert calls == calls_ref, f"model calls: oracle {calls} vs upstream {calls_ref}"
    assert torch.equal(got, want)


def test_stand_in_model_sees_every_input():
    """The stand-in UNet must react to each conditioning input, or the comparison above would not see a routing error."""
    case = dict(ed.ENTRY_CASES[2])
    base, _ = ed.oracle_run(case)
    for key in ("i2i",):
        alt = dict(case)
        alt[key] = not case[key]
        assert not torch.equal(ed.oracle_run(alt)[0], base)
    x, cond, _ = ed.inputs(case)
    f = ed.make_model([])
    sig = torch.ones(2)
    ref = f(x, sig, cond=cond)
    for k in ("crossattn", "vector"):
        c2 = dict(cond)
        c2[k] = cond[k].flip(0)
        assert not torch.equal(f(x, sig, cond=c2), ref), k
    c2 = dict(cond)
    c2["c_concat"] = [cond["c_concat"][0].flip(-1)]
    assert not torch.equal(f(x, sig, cond=c2), ref)
    assert not torch.equal(f(x, sig * 2, cond=cond), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ed.ENTRY_CASES, ids=IDS)
def test_plugin_entry_path_matches_oracle(plugin, cuda, case, golden_entry):
    got, calls = ed.drive(plugin, case, cuda, ed.plugin_regions(plugin))
    want, calls_o = ed.oracle_run(case)
    assert calls == calls_o, f"model calls: plugin {calls} vs oracle {calls_o}"
    assert got.is_cuda
    assert torch.equal(got.cpu(), want)
    assert np.array_equal(got.cpu().numpy(), golden_entry[case["name"] + "/outs"])
