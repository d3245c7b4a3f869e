import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import icp_rust_amd as I
from icp_rust_amd import synth
src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
res = None
for _ in range(3): res = icp.estimate(d_src, I.Transform(), 20)
ts = []
for _ in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter(); res = icp.estimate(d_src, I.Transform(), 20); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 20)
ts.sort()
print(f"estimate(20): {1e3 * ts[len(ts)//2]:.4f} ms/step; run-ahead (hits, misses) {I.run_ahead_counters(icp)}; paths {I.gn_path_counters(icp)}")
p = res.pose if hasattr(res, 'pose') else res
print("pose", p)
