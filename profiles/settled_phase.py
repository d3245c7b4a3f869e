"""What an outer iteration costs once a registration has settled (converging pair, 1M points: the inner loop applies no
update any more, an outer iteration is one search and one evaluation), with and without certified matches:
    python3 profiles/settled_phase.py            (ICP_NN_NO_CERT=1 for the other leg)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import icp_rust_amd as I
from icp_rust_amd import synth

src, dst, _ = synth.converging_pair(1_000_000, 1_000_000)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
icp.estimate(d_src, I.Transform(), 20)
res = {}
for iters in (20, 60):
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T, idx, inner = icp.estimate(d_src, I.Transform(), iters, return_info=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    res[iters] = sorted(ts)[len(ts) // 2]
cs = I.nn_cert_counters(icp)
print(f"NO_CERT={os.environ.get('ICP_NN_NO_CERT')}: estimate(20) {1e3 * res[20]:.2f} ms, estimate(60) {1e3 * res[60]:.2f} ms -> "
      f"{1e6 * (res[60] - res[20]) / 40:.1f} us per settled outer iteration; certified searches {cs[0]}, failed in the last {cs[1]}")
