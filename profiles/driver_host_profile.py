import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import icp_rust_amd as I
from icp_rust_amd import synth
from icp_rust_amd.dist import BlockShardedIcp, HipStages, LocalComm
n = 1_000_000
src, dst = synth.synthetic_pair(n, n)
d_src = torch.from_numpy(src).cuda(); d_dst = torch.from_numpy(dst).cuda()
icp2 = I.Icp3d(d_dst)
drv = BlockShardedIcp({0: HipStages(icp2)}, n, 1, LocalComm(1))
srt, perms = drv.sort_source(d_src, I.Transform())
local = drv.take_source(srt)
drv.estimate(local, I.Transform(), 20)
pr = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.perf_counter()
pr.enable()
drv.estimate(local, I.Transform(), 20)
pr.disable()
torch.cuda.synchronize(); print("per iteration ms", 1e3 * (time.perf_counter() - t0) / 20)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
