import os
import sys

import pytest

# PyTorch-ROCm brings its own copy of the HIP runtime and libpt_hip.so is linked against the system's: a process that
# uses both must load torch's first (the library then binds to the runtime already in memory). Loaded the other way
# round, torch finds a runtime it was not built with and reports "No HIP GPUs are available". Several GPU tests hand
# torch tensors to the library, so every test session imports torch before anything else.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle_lib import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def dev():
    """One pt_handle for the whole session. There is no CPU fallback: this raises without a GPU."""
    from path_tracer_amd import Device
    d = Device(0)
    yield d
    d.close()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
