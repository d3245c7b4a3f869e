"""True multi-process ranks on ONE GPU: the inner loops run as one launch per process and exchange
through hipIpc-mapped inboxes (include/icp_mi355x.h section 5b); the pose must be one handle's, bit for bit."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "helpers", "two_process_loop.py")

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(world, *args):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, HELPER, *[str(a) for a in args]], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    return [p.returncode for p in procs], outs


# (two processes: three on one GPU did not run side by side on the test box -- the third queue waited for a slice and the
# launches, which wait for each other, gave up after their 250 ms: a property of sharing ONE GPU, not of the protocol)
@pytest.mark.parametrize("world,n,m,kind", [(2, 150_000, 120_000, "independent"), (2, 200_000, 200_000, "converging")])
def test_processes_sharing_a_gpu_exchange_through_ipc_inboxes(world, n, m, kind):
    # The launches of the two processes wait for each other, so both must be RUNNING on the one GPU at once; the platform
    # does not promise that to two processes (the second queue may sit out a time slice), and a launch gives up after its
    # bounded wait (ICP_HIP_ERROR from icp_shard_loop_wait) rather than hang.  That -- and only that -- is retried, and
    # skipped if the box never runs the two side by side; wrong bits or any other error fail at once.
    for attempt in range(3):
        rcs, outs = run_ranks(world, n, m, 6, kind)
        if all(rc == 0 for rc in rcs):
            assert "pose equals one handle's: True" in outs[0], outs[0]
            return
        text = "\n".join(outs)
        gave_up = "icp_shard_loop_wait: HIP error" in text
        assert gave_up and "pose equals one handle's: False" not in text, text
    pytest.skip("two processes were not scheduled side by side on this GPU in three attempts (bounded waits gave up)")
