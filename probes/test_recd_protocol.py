"""CPU check of the request / wait / read schedule of the dripped-epilogue record conv (probes/csrc/vae_conv_recd.hip; PROBES twin only since round 6): operand DMA, slot
stores and residual loads share ONE in-order vmcnt; the waits in front of the phase barriers count lower bounds of what the slots
issued.  probes/recd_protocol_sim.py re-states the kernel's loops (slot trip + plain trips, three pieces per phase, the two wave
groups passing a phase's barrier half a step apart) on the adversarial memory model of tools/rec2_protocol_sim.py."""
import importlib.util
import inspect
import itertools
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sim():
    spec = importlib.util.spec_from_file_location("recd_protocol_sim", os.path.join(ROOT, "probes", "recd_protocol_sim.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_protocol_holds_for_every_output_kind_depth_and_item_count():
    sim = _sim()
    for NK in (8, 10, 16):
        for items in (1, 2, 3):
            for flags in itertools.product((False, True), repeat=4):      # fp32 output, record output, residual, rows inside the image
                if flags[0] or flags[1]:
                    for slack in (0, 6):                                    # border cells: more stores than the lower bound
                        assert sim.sim(NK, items, flags, slack) == [], (NK, items, flags, slack)


def test_the_model_rejects_looser_waits():
    """one more instruction allowed in flight than a slot is known to have issued, or the input-stage wait at the wrong phase: an operand
    may be read before it has landed -- the model is able to see the failures it is there to exclude"""
    sim = _sim()
    full = (True, True, True, True)
    sim.BIG = 17
    assert sim.sim(8, 3, full)
    sim.BIG = 15
    sim.SMALL = 3
    assert sim.sim(8, 3, (False, True, False, True))
    sim.SMALL = 2
    src = inspect.getsource(sim.sim)
    base = "return 5 if (dy == 1 and k + 1 < NK) else 0"
    assert base in src
    for worse in ("return 6 if (dy == 1 and k + 1 < NK) else 0", "return 5 if (dy >= 1 and k + 1 < NK) else 0"):
        ns = {}
        exec(src.replace(base, worse), dict(vars(sim)), ns)
        assert ns["sim"](8, 3, full), worse


def test_source_constants_match_the_kernel():
    """the lower bounds and the slot map of the model are the kernel's"""
    sim = _sim()
    hip = open(os.path.join(ROOT, "probes", "csrc", "vae_conv_recd.hip")).read()
    assert "if (big) { MDT_VMCNT(15); counted = true; }" in hip and "MDT_VMCNT(2); counted = true;" in hip and "MDT_VMCNT(5);" in hip
    assert "constexpr int D_TK = 4;" in hip and sim.D_TK == 4
    assert "return p % 3 == 0 ? -1 : 2 * (p / 3) + p % 3 - 1;" in hip
    assert [sim.slot_kind(p) for p in range(12)] == [0, 1, 2] * 4
