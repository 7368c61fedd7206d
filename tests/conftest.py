import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tiny_scene():
    """BASELINE.json configs[0]: 3 views 640x480 (synthetic, deterministic)."""
    from hpmvs_amd import synth
    return synth.make_scene(3, 640, 480, n_waves=24)


@pytest.fixture(scope="session")
def tiny_seeds(tiny_scene):
    """~200 seeds.  start_level=2 (not the reference's 4): a 640x480 level-4 image is 40x30 px, too
    small for a 7x7 window plus the 3 px margin almost everywhere."""
    from hpmvs_amd import synth
    return synth.make_seeds(tiny_scene, 200, start_level=2)


@pytest.fixture(scope="session")
def oracle_scene(tiny_scene):
    from oracle import oracle as orc
    orc.build()
    return orc.OracleScene(tiny_scene)


@pytest.fixture(scope="session")
def gpu_scene(tiny_scene):
    from hpmvs_amd import api
    if api.device_count() < 1:
        pytest.fail("no HIP device: -m gpu tests need the MI355X box (no CPU fallback exists)")
    sc = api.Scene(tiny_scene, device=0)
    yield sc
    sc.close()
