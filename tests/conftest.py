import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.exists("/dev/kfd"):
    # No GPU here (set before torch / libgomp load): idle OpenMP workers sleep instead of spinning.  Measured on the 8-core build container
    # (round 4, whole `-m "not gpu"` suite): quiet box 3 min 49 s passive / 2 min 31 s spinning; with six busy neighbour processes
    # 5 min 06 s passive / 13 min 46 s spinning — the 20-40-minute runs of earlier rounds were spinning workers fighting the neighbours.
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    os.environ.setdefault("GOMP_SPINCOUNT", "0")
for p in (os.path.join(ROOT, "e4t-diffusion_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    # The CPU suite runs small models; on a shared 8-core container its wall time tracks the neighbours' load (3 min 20 s on a quiet
    # box, 23-42 min on a busy one, same code): oversubscribed intra-op thread pools spin against each other and against the 2-rank
    # gloo children.  Four threads per process are enough for these sizes and leave the box responsive.  (On the GPU box the CPU
    # oracle of the full-size cases keeps every core.)
    import torch
    if not torch.cuda.is_available():
        torch.set_num_threads(min(4, os.cpu_count() or 4))
        os.environ.setdefault("OMP_NUM_THREADS", "4")


@pytest.fixture(scope="session")
def hip_env():
    """(hip backend, emu backend, device, ops module) on the GPU box; loud failure if the extension is missing."""
    import torch
    from e4t import ops
    from emu_backend import EmuBackend

    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    hip = ops.HipBackend()   # raises if libe4t_hip.so is missing: no fallback
    return hip, EmuBackend(), torch.device("cuda:0"), ops
