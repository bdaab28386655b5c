import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.library()
    return binding


@pytest.fixture(scope="session")
def bflib():
    """The product library through its C ABI; GPU tests fail loudly when it cannot compute."""
    from ogl_beamforming_amd import lib
    return lib


@pytest.fixture
def hooks(bflib):
    """Test / measurement hooks of the library (beamformer_hip_set_hook), all switched off again after the test."""
    used = []

    class Setter:
        def set(self, name, value="1"):
            bflib.set_hook(name, value)
            used.append(name)

        def clear(self, name):
            bflib.set_hook(name, None)

    yield Setter()
    for name in used:
        bflib.set_hook(name, None)
