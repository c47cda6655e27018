"""-m gpu: the collective path on real RCCL (one rank: the build has no multi-GPU box; N > 1 is covered on CPU by the 2-rank
gloo tests in test_ddp_gloo.py).  Launched exactly the way the driver launches bench.py for N > 1."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_one_rank_rccl_step_equals_no_comm_step(hip_env):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "rccl_one_rank.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
