"""What the N-rank orchestration costs, measured with VIRTUAL ranks (every rank on cuda:0).  Round 4: an outer
iteration is, per rank, one search launch + one inner-loop launch (gn_loop.hip: k_gn_loop_shard; the launches of ranks
that share a device are fused into one); the N searches of the virtual ranks run back to back on one stream, so
(time at N) - (time at 1) is mostly N small searches in a row plus the exchange inside the loop launch.  On N real
GPUs the per-rank kernels run side by side (DESIGN.md section 7 turns this into the expected scaling curve).
"per outer iteration" = a whole estimate(20) from HOST buffers / 20 (upload, sort, shard and index read-back
included); "marginal" = (estimate(40) - estimate(20)) / 20, the per-call work cancelled."""
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
        marginal = 1e3 * (dt40 - dt) / 20
        evals = int(inn.sum()) + 20
        print(f"icp_create_multi, {W} virtual ranks on one GPU: {1e3 * dt / 20:.3f} ms per outer iteration "
              f"({evals} evaluations; counters sharded/replicated {mu.counters()}, loop launches/served/handbacks {mu.loop_counters()})")
        print(f"    marginal outer iteration (40 against 20 iterations, per-call work cancelled): {marginal:.3f} ms")
        mu.close()


if __name__ == "__main__":
    main()
