import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_rust_amd as I
from icp_rust_amd import synth
pk = synth.synthetic_scan3d_packets(150)
s3 = synth.remove_invalid_values(pk[:75]); d3 = synth.remove_invalid_values(pk[75:150])
ds3 = torch.from_numpy(s3).cuda(); dd3 = torch.from_numpy(d3).cuda()
def t(f, k=20):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(k): f()
    return 1e3 * (time.perf_counter() - t0) / k
icp = I.Icp3d(d3)
print("estimate host src  ", t(lambda: icp.estimate(s3, I.Transform(), 20)))
print("estimate device src", t(lambda: icp.estimate(ds3, I.Transform(), 20)))
print("estimate host src, 1 iter  ", t(lambda: icp.estimate(s3, I.Transform(), 1)))
print("estimate device src, 1 iter", t(lambda: icp.estimate(ds3, I.Transform(), 1)))
print("new host dst  ", t(lambda: I.Icp3d(d3).close()))
print("new device dst", t(lambda: I.Icp3d(dd3).close()))
x = torch.empty_like(ds3)
def cp():
    x.copy_(torch.from_numpy(s3)); torch.cuda.synchronize()
print("torch H2D pageable 690KB", t(cp))
ps = torch.from_numpy(s3).pin_memory()
def cp2():
    x.copy_(ps); torch.cuda.synchronize()
print("torch H2D pinned 690KB", t(cp2))
