"""The weak-scaling configuration of 8 ranks (8 x 1M source points against the 1M-point target) on VIRTUAL ranks of one GPU:
the streamed one-launch loop (K up to 64 pairs per thread) against the stage calls (ICP_NO_GN_LOOP=1), and both against
ONE handle, bit for bit.  usage: python3 profiles/multi_weak_8m.py [points, default 8M] [ranks, default 8]"""
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
one.close()
print(f"one handle, {n} points: {1e3 * t_one / 6:.3f} ms per outer iteration (host buffers), inner {list(map(int, inner1))}")
mu = I.IcpMulti(dst, [0] * W)
mu.estimate(src, init, 2)
t0 = time.perf_counter()
T, idx, inner = mu.estimate(src, init, 6, return_info=True)
dt = time.perf_counter() - t0
ok = np.array_equal(T.as_array(), T1.as_array()) and np.array_equal(inner, inner1) and np.array_equal(idx, idx1)
print(f"{W} virtual ranks ({'stage calls' if os.environ.get('ICP_NO_GN_LOOP') else 'one-launch loop where it applies'}): "
      f"{1e3 * dt / 6:.3f} ms per outer iteration; same bits as one handle: {ok}; loop (launches, served, handbacks) {mu.loop_counters()}; "
      f"counters {mu.counters()}")
mu.close()
sys.exit(0 if ok else 3)
