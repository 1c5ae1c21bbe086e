"""CPU check of the hand-counted LDS-DMA protocol of the two-blocks-per-CU record conv kernels (csrc/vae_conv_rec2.hip): the
request / counted-wait / read schedule is re-stated in tools/rec2_protocol_sim.py and run against an adversarial memory model
(pieces land as late as the vmcnt waits allow; ring slots and input stages are re-used across steps, K-steps and items)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sim():
    spec = importlib.util.spec_from_file_location("rec2_protocol_sim", os.path.join(ROOT, "tools", "rec2_protocol_sim.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_protocol_holds_for_every_depth_and_item_count():
    sim = _sim()
    for NK in (2, 4, 6, 8, 16, 32):
        for items in (1, 2, 3, 4):
            assert sim.sim_conv3x3(NK, items) == []
            assert sim.sim_upconv(NK, items) == []


def test_the_model_rejects_a_wait_that_is_one_piece_too_loose():
    """the counted waits are tight: the same schedule with ONE more piece allowed in flight at any step position reads an operand
    that may not have landed -- so the model is able to see the failure it is there to exclude"""
    import inspect
    sim = _sim()
    src = inspect.getsource(sim.sim_conv3x3)
    base = "N_OF_S = [2, 3, 4, 4, 4, 4, 3, 2, 2]"
    assert base in src
    for pos in range(9):
        vals = [2, 3, 4, 4, 4, 4, 3, 2, 2]
        vals[pos] += 1
        ns = {}
        exec(src.replace(base, f"N_OF_S = {vals}"), dict(vars(sim)), ns)
        assert ns["sim_conv3x3"](8, 3), f"loosening step position {pos} went unnoticed"
    src = inspect.getsource(sim.sim_upconv)
    base = "N_OF_E = [6, 7, 8, 9, 8, 7, 6, 6]"
    assert base in src
    for pos in range(8):
        vals = [6, 7, 8, 9, 8, 7, 6, 6]
        vals[pos] += 1
        ns = {}
        exec(src.replace(base, f"N_OF_E = {vals}"), dict(vars(sim)), ns)
        assert ns["sim_upconv"](8, 3), f"loosening step position {pos} went unnoticed"
