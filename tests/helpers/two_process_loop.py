"""One rank of tests/test_gpu_ipc.py: two (or more) PROCESSES on one GPU, the source cloud sharded by reduction-tree
block, every inner loop one launch per process (gn_loop.hip: k_gn_loop_shard) whose workgroups exchange histograms,
candidates and block sums through hipIpc-mapped inboxes -- no host staging, no collective on the per-iteration path.
gloo only carries the rendezvous (the IPC handles at connect time) and whatever evaluation a launch hands back."""
import datetime
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import icp_rust_amd as I  # noqa: E402
from icp_rust_amd import synth  # noqa: E402
from icp_rust_amd.dist import BlockShardedIcp, HipStages, TorchComm  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n, m, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    kind = sys.argv[4] if len(sys.argv) > 4 else "independent"
    transport = sys.argv[5] if len(sys.argv) > 5 else "auto"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
    src, dst = (synth.converging_pair(n, m)[:2] if kind == "converging" else synth.synthetic_pair(n, m))
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    icp = I.Icp3d(d_dst)
    driver = BlockShardedIcp({rank: HipStages(icp)}, n, world, TorchComm(rank, world))
    got = driver.connect_loop(transport=transport)
    if got is None:  # (e.g. a runtime that does not export fine-grained allocations: agreed on by every rank)
        print(f"rank {rank}: transport {transport} not available here", flush=True)
        dist.barrier()
        icp.close()
        sys.exit(5)
    init = I.Transform([0.01, -0.02, 0.001])
    T, inner, _ = driver.estimate_full(d_src, init, iters)
    T2, inner2, _ = driver.estimate_full(d_src, init, iters)  # (generations and parities carry over)
    assert np.array_equal(T.as_array(), T2.as_array()) and np.array_equal(inner, inner2)
    c = driver.counters
    if os.environ.get("ICP_DIST_TEST_WITHHOLD"):  # (a forced give-up: every rank must have landed on the stage calls, once)
        assert c.get("loop_gave_up", 0) == 1 and driver._loop is None, c
        print(f"rank {rank}: forced give-up: every rank restarted the call through the stage calls; counters {c}", flush=True)
    elif c.get("loop_gave_up", 0):  # (the two processes were not scheduled side by side: the stage calls served -- same bits)
        print(f"rank {rank}: icp_shard_loop_wait: HIP error (a launch gave up waiting)", flush=True)
    else:
        # every outer iteration through the inboxes: a one-launch inner loop, or the pipelined evaluation (round 6)
        assert c["loop_launches"] + c.get("pipe_iterations", 0) >= iters and c["loop_served"] + 2 * c.get("pipe_iterations", 0) >= iters, c
        if kind != "converging":
            assert c.get("pipe_iterations", 0) >= 2 * (iters - 4), c  # (both calls: all but their first iterations)
    ok = 1
    if rank == 0:
        one = I.Icp3d(d_dst)
        T1, inner1 = one.estimate(d_src, init, iters, return_info="inner")
        same = np.array_equal(T.as_array(), T1.as_array()) and np.array_equal(inner, inner1)
        print(f"rank 0: transport {got}: pose equals one handle's: {same}; inner {inner.tolist()}; counters {c}", flush=True)
        ok = 1 if same else 0
    t = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    dist.barrier()
    driver.disconnect_loop()
    icp.close()
    forced = bool(os.environ.get("ICP_DIST_TEST_WITHHOLD"))
    sys.exit((4 if (c.get("loop_gave_up", 0) and not forced) else 0) if int(t.item()) == 1 else 3)


if __name__ == "__main__":
    main()
