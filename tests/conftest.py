import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "e4t-diffusion_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def hip_env():
    """(hip backend, emu backend, device, ops module) on the GPU box; loud failure if the extension is missing."""
    import torch
    from e4t import ops
    from emu_backend import EmuBackend

    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    hip = ops.HipBackend()   # raises if libe4t_hip.so is missing: no fallback
    return hip, EmuBackend(), torch.device("cuda:0"), ops
