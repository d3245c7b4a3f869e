"""Two INDEPENDENT processes on one GPU, each with its own handle: the one-launch inner loop waits at
grid barriers for workgroups that another tenant's kernels may be keeping off the CUs.  Round 4 waited 250 ms before
handing such a loop back and then switched the path off for the life of the handle; now the wait ends 2 ms after the last
workgroup it saw arrive, the handle steps its next 64 inner loops from the host and tries again.  Checked here: every call
of either tenant returns the bits of that tenant's solo run, and no call takes anywhere near the old cliff."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "helpers", "independent_handle.py")

pytestmark = pytest.mark.gpu


def run_tenants(k, calls, workload="frame"):
    with tempfile.TemporaryDirectory() as sync:
        procs = [subprocess.Popen([sys.executable, HELPER, str(calls), str(i + 1), sync, str(k), workload], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
                 for i in range(k)]
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


def parse(out):
    line = [ln for ln in out.splitlines() if ln.startswith("tenant ")][-1]
    worst = float(line.split("worst ")[1].split(" ms")[0])
    p98 = float(line.split("p98 ")[1].split(" ms")[0])
    pose = line.split("pose ")[1]
    return line, worst, p98, pose


def test_two_tenants_keep_their_bits_and_their_latency():
    solo_rc, solo_out = run_tenants(1, 20)
    assert solo_rc == [0], solo_out
    rcs, outs = run_tenants(2, 200)
    assert rcs == [0, 0], "\n".join(outs)
    for i, out in enumerate(outs):
        line, worst, p98, pose = parse(out)
        assert "same bits every call: True" in line, line
        # 98 of 100 calls within 10 ms (ten times a frame's registration, a twenty-fifth of the round-4 cliff; a
        # co-tenant's own kernels are in this number too), and no call near the cliff: the worst call also carries
        # whatever the host's scheduler did to either process (a measured run: 1.9 ms / 2.5 ms worst, one box 11.9 ms)
        assert p98 < 10.0, line
        assert worst < 100.0, line
    # tenant 1 of the pair ran the registration the solo tenant ran: the same pose, bit for bit
    assert parse(outs[0])[3] == parse(solo_out[0])[3], (outs[0], solo_out[0])


def test_two_tenants_whose_every_iteration_is_a_one_launch_loop():
    """the frame above takes the host-stepped bet (one update per iteration) and never launches
    k_gn_loop in the steady state, so its grid barriers never contend.  Here both tenants register a CONVERGING pair: every
    outer iteration is a one-launch inner loop of several evaluations in both processes at once -- 256 workgroups each that
    wait for each other at grid barriers on one GPU.  A launch that is not resident gives up after a bounded wait and the
    handle steps from the host for a while: bits never change, and no call takes anywhere near the round-4 cliff."""
    solo_rc, solo_out = run_tenants(1, 10, "converging")
    assert solo_rc == [0], solo_out
    rcs, outs = run_tenants(2, 60, "converging")
    assert rcs == [0, 0], "\n".join(outs)
    for out in outs:
        line, worst, p98, pose = parse(out)
        assert "same bits every call: True" in line, line
        launches = int(line.split("loop (launches, evals, handbacks) (")[1].split(",")[0])
        assert launches >= 60, line  # the loops really were one-launch loops (a tenant that times out steps 64 from the host, then tries again)
        assert worst < 100.0, line
        print(line)
    assert parse(outs[0])[3] == parse(solo_out[0])[3], (outs[0], solo_out[0])
