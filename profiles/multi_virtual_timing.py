"""What the N-rank orchestration costs, measured with VIRTUAL ranks (every rank on cuda:0, one stream,
lockstep): the GPU work is the one-GPU work split N ways and run back to back, so (time at N) - (time at 1)
is the price of the extra launches and exchange kernels -- the part that does NOT shrink with more GPUs.
On N real GPUs the per-rank kernels run side by side; the expected step is
    max over ranks (search + evaluations of n/N points) + that fixed price / 1 (it is per rank, in parallel)
(DESIGN.md section 7 turns this into the expected scaling curve)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import icp_rust_amd as I  # noqa: E402
from icp_rust_amd import synth  # noqa: E402


def main():
    n = m = 1_000_000
    src, dst = synth.synthetic_pair(n, m)
    one = I.Icp3d(dst)
    one.estimate(src, I.Transform(), 5)
    t0 = time.perf_counter()
    T1, _, inner = one.estimate(src, I.Transform(), 20, return_info=True)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    one.estimate(src, I.Transform(), 40)
    t2 = time.perf_counter() - t0
    print(f"one handle (host buffers): {1e3 * t1 / 20:.3f} ms per outer iteration; marginal {1e3 * (t2 - t1) / 20:.3f} ms")
    for W in (1, 2, 4, 8):
        mu = I.IcpMulti(dst, [0] * W)
        mu.estimate(src, I.Transform(), 5)
        t0 = time.perf_counter()
        T, _, inn = mu.estimate(src, I.Transform(), 20, return_info=True)
        dt = time.perf_counter() - t0
        assert np.array_equal(T.as_array(), T1.as_array())
        t0 = time.perf_counter()
        mu.estimate(src, I.Transform(), 40)
        dt40 = time.perf_counter() - t0
        print(f"    marginal outer iteration (40 against 20 iterations, per-call work cancelled): {1e3 * (dt40 - dt) / 20:.3f} ms")
        evals = int(inn.sum()) + 20
        print(f"icp_create_multi, {W} virtual ranks on one GPU: {1e3 * dt / 20:.3f} ms per outer iteration "
              f"({evals} evaluations; counters sharded/replicated {mu.counters()}, loop launches/served/handbacks {mu.loop_counters()})")
        mu.close()


if __name__ == "__main__":
    main()
