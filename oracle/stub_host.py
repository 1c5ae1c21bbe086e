"""Moved to hostsim/stub_host.py (round 4): the A1111 host stand-in drives the product, it is not part of the checker.
This alias keeps `from oracle import stub_host` working (ONE module object: the stub keeps install state)."""
import sys

from hostsim import stub_host as _moved

sys.modules[__name__] = _moved
