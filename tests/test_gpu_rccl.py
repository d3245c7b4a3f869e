"""RCCL itself (backend "nccl") with the tensors the sharded driver hands it -- a one-rank world, all a one-GPU box can
form.  Everything else about N ranks runs over gloo (tests/test_dist_gloo.py, test_gpu_ipc.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_the_drivers_collectives_run_on_rccl():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_world1.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "every collective of the sharded driver ran" in p.stdout
