"""The weak-scaling configuration of 8 ranks (8 x 1M source points against the 1M-point target) on VIRTUAL ranks of one GPU
(round 6: the tree grows with the cloud, every rank owns 256 blocks and the steady state runs through the pipelined
evaluation, csrc/pipe.hip) against ONE handle, bit for bit.  Calls are from HOST buffers (200 MB up, 32 MB back per call):
"marginal" = (estimate(26) - estimate(6)) / 20 cancels the per-call work.
usage: python3 profiles/multi_weak_8m.py [points, default 8M] [ranks, default 8]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import icp_rust_amd as I  # noqa: E402
from icp_rust_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8 * 1024 * 1024
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
src, dst = synth.synthetic_pair(n, 1_000_000)
init = I.Transform([0.01, -0.02, 0.001])
one = I.Icp3d(dst)
one.estimate(src, init, 2)
t0 = time.perf_counter()
T1, idx1, inner1 = one.estimate(src, init, 6, return_info=True)
t_one = time.perf_counter() - t0
t0 = time.perf_counter()
one.estimate(src, init, 26)
t_one26 = time.perf_counter() - t0
one.close()
print(f"one handle, {n} points: {1e3 * t_one / 6:.3f} ms per outer iteration (host buffers), marginal {1e3 * (t_one26 - t_one) / 20:.3f} ms; "
      f"inner {list(map(int, inner1))}")
mu = I.IcpMulti(dst, [0] * W)
mu.estimate(src, init, 2)
t0 = time.perf_counter()
T, idx, inner = mu.estimate(src, init, 6, return_info=True)
dt = time.perf_counter() - t0
ok = np.array_equal(T.as_array(), T1.as_array()) and np.array_equal(inner, inner1) and np.array_equal(idx, idx1)
t0 = time.perf_counter()
mu.estimate(src, init, 26)
dt26 = time.perf_counter() - t0
print(f"{W} virtual ranks ({'stage calls' if os.environ.get('ICP_NO_GN_LOOP') else 'pipelined evaluation where it applies'}): "
      f"{1e3 * dt / 6:.3f} ms per outer iteration, marginal {1e3 * (dt26 - dt) / 20:.3f} ms; same bits as one handle: {ok}; "
      f"pipelined iterations {mu.pipe_iterations()}; loop (launches, served, handbacks) {mu.loop_counters()}; counters {mu.counters()}")
mu.close()
sys.exit(0 if ok else 3)
