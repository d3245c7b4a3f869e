"""What a call costs beyond its steady-state iterations: estimate(src, I, k) on the 1M pair for k = 1, 2, 3, 5, 10, 20, 40
(resident inputs, median of 9), the differences between them, and the same through the C entry point alone (no Python
wrapper objects per call)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import icp_rust_amd as I
from icp_rust_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
src, dst = synth.synthetic_pair(n, n)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
T0 = I.Transform()
for _ in range(3):
    icp.estimate(d_src, T0, 20)
torch.cuda.synchronize()
prev = None
for k in (1, 2, 3, 5, 10, 20, 40):
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        icp.estimate(d_src, T0, k)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = 1e3 * ts[len(ts) // 2]
    extra = "" if prev is None else f"  (+{(med - prev[1]) / (k - prev[0]):.4f} ms per added iteration)"
    print(f"estimate(k = {k:2d}): {med:.3f} ms{extra}", flush=True)
    prev = (k, med)
# the C call alone
lib = I.lib()
o = I.Transform()
inner = np.zeros(64, dtype=np.uint32)
ts = []
for _ in range(9):
    t0 = time.perf_counter()
    lib.icp_estimate_device(icp._h, C.c_void_p(d_src.data_ptr()), n, C.byref(T0.pose), 20, C.byref(o.pose), None, C.c_void_p(inner.ctypes.data))
    ts.append(time.perf_counter() - t0)
ts.sort()
print(f"icp_estimate_device(k = 20) through ctypes alone: {1e3 * ts[len(ts) // 2]:.3f} ms")
