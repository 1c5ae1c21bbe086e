"""Test configuration.  `-m "not gpu"` runs here (no GPU): oracle vs golden vectors / vs the upstream reference, host
logic, C-ABI symbol checks.  `-m gpu` runs on an MI355X: the HIP path through the C ABI vs the oracle and the goldens."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cases():
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_blend():
    return np.load(os.path.join(GOLDEN, "blend.npz"))


@pytest.fixture(scope="session")
def golden_maps():
    return np.load(os.path.join(GOLDEN, "maps.npz"))


@pytest.fixture(scope="session")
def golden_vae():
    return np.load(os.path.join(GOLDEN, "vae.npz"))


@pytest.fixture(scope="session")
def built_lib():
    """libmdtile.so, (re)built in-tree if the sources changed (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    ge.build()
    from hostsim import stub_host
    return stub_host.load_plugin().engine


@pytest.fixture(scope="session")
def plugin(built_lib):
    from hostsim import stub_host
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    stub_host.install(dev)
    stub_host.set_device(dev)
    return stub_host.load_plugin()


@pytest.fixture(scope="session")
def cuda(plugin):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
