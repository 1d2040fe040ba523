import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_plane_scene():
    """cfg1-like: 3 views 320x240 of one textured plane, 30 degree arc."""
    from mvskit_amd import synth

    return synth.make_scene(nviews=3, W=320, H=240, arc_deg=30.0, radius=4.0, kind="plane")


@pytest.fixture(scope="session")
def small_multi_scene():
    """cfg2-like at toy size: 5 views 384x216, planes + sphere, 60 degree arc."""
    from mvskit_amd import synth

    return synth.make_scene(nviews=5, W=384, H=216, arc_deg=60.0, radius=4.0, kind="multi")
