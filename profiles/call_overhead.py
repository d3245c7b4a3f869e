"""Diagnostic: wall time of whole estimate calls of 1, 2, 3, 20 outer iterations on a resident 1M x 1M
pair -- what a call costs beyond its iterations (snapshot of the source cloud, cold first search)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import icp_rust_amd as I
from icp_rust_amd import synth
n = m = 1_000_000
src, dst = synth.synthetic_pair(n, m)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
icp.estimate(d_src, I.Transform(), 3)
for iters in (1, 2, 3, 5, 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        icp.estimate(d_src, I.Transform(), iters)
    torch.cuda.synchronize()
    print(f"estimate(.., {iters:2d}): {(time.perf_counter() - t0) / 10 * 1e6:8.1f} us per call")
