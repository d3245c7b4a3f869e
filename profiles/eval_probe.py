"""One Gauss-Newton evaluation at a time on the pairs of the benchmark cloud (icp_weighted_gn_step_device): the kernels
of an evaluation ALONE on the chip -- no search beside them -- so that `rocprofv3 --kernel-trace --stats` of this
script gives their stand-alone durations and the wall time per call the whole chain including the host's wait.
usage: eval_probe.py [n_points] [calls]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "/root/repo")
import icp_rust_amd as I
from icp_rust_amd import synth
from icp_rust_amd._lib import lib, Pose

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
src, dst = synth.synthetic_pair(n, n)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
d_a = torch.empty((n, 2), dtype=torch.float64, device="cuda")
d_b = torch.empty((n, 2), dtype=torch.float64, device="cuda")
icp.correspond_device(d_src, I.Transform(), d_a, d_b)
torch.cuda.synchronize()
T = I.Transform()._pose if hasattr(I.Transform(), "_pose") else None
pose = Pose(1.0, 0.0, 0.0, 1.0, 0.0, 0.0)
delta = (C.c_double * 3)()
herr = C.c_double()


def step(kind=0):
    rc = lib().icp_weighted_gn_step_device(icp._h, C.c_void_p(d_a.data_ptr()), C.c_void_p(d_b.data_ptr()), n, C.byref(pose),
                                           kind, delta, C.byref(herr))
    assert rc == 0, rc


for _ in range(5):
    step()
ts = []
for _ in range(calls):
    t0 = time.perf_counter()
    step()
    ts.append(time.perf_counter() - t0)
ts.sort()
print(f"n = {n}: one evaluation (launch .. result seen by the host): median {1e6 * ts[len(ts) // 2]:.1f} us, min {1e6 * ts[0]:.1f} us; "
      f"delta {list(delta)} huber {herr.value:.6f}; paths {I.gn_path_counters(icp)}")
