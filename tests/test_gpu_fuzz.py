"""A slice of the device fuzz (tests/tools/fuzz/gpu1.py) in the GPU suite: random sizes, bit depths, presets and lookahead options,
the whole lookahead on the device against the real reference build.  The fuzzer itself runs hundreds of configurations per minute
on a GPU box; here a fixed seed keeps the suite deterministic."""
import importlib.util
import os

import pytest

from oracle import refharness

pytestmark = pytest.mark.gpu


def _fuzzer():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "fuzz", "gpu1.py")
    spec = importlib.util.spec_from_file_location("gpu_fuzz1", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.skipif(not refharness.available(8) or not refharness.available(10), reason="oracle/_ref did not travel")
@pytest.mark.parametrize("seed,n,big", [(101, 60, False), (102, 6, True)])
def test_random_configurations_match_the_reference(seed, n, big):
    assert _fuzzer().run(seed, n, big=big, verbose=False) == 0
