"""Diagnostic: what Icp3d::new (icp_create_device) + drop cost per frame, by target size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import icp_rust_amd as I
from icp_rust_amd import synth
for m in (28_000, 200_000, 1_000_000):
    src, dst = synth.synthetic_pair(1000, m)
    d_dst = torch.from_numpy(dst).cuda()
    for rep in range(2):
        icp = I.Icp3d(d_dst); icp.synchronize(); icp.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(10):
        icp = I.Icp3d(d_dst)
        icp.synchronize()
        icp.close()
    dt = (time.perf_counter() - t0) / 10
    print(f"m = {m:8d}: Icp3d::new + drop {dt * 1e3:7.3f} ms")
