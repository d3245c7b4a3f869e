import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, icp_rust_amd as I
from icp_rust_amd import synth
src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
icp.estimate(d_src, I.Transform(), 3)
for _ in range(20):
    icp.estimate(d_src, I.Transform(), 1)
torch.cuda.synchronize()
