"""Cost of the block-sharded driver itself (dist.BlockShardedIcp: Python + stage calls, one host wait per
stage) with ONE rank and no collective, against the library's own loop on the same pair."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import icp_rust_amd as I
from icp_rust_amd import synth
from icp_rust_amd.dist import BlockShardedIcp, HipStages, LocalComm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
src, dst = synth.synthetic_pair(n, n)
d_src = torch.from_numpy(src).cuda(); d_dst = torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    T = icp.estimate(d_src, I.Transform(), 20)
    torch.cuda.synchronize(); t_lib = (time.perf_counter() - t0) / 20
icp2 = I.Icp3d(d_dst)
drv = BlockShardedIcp({0: HipStages(icp2)}, n, 1, LocalComm(1))
srt, perms = drv.sort_source(d_src, I.Transform())  # (the fold order of the library's call: same bits)
local = drv.take_source(srt)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    T2, inner = drv.estimate(local, I.Transform(), 20)
    torch.cuda.synchronize(); t_drv = (time.perf_counter() - t0) / 20
print(f"{n} points: library loop {1e3 * t_lib:.3f} ms per outer iteration; block-sharded driver, 1 rank, no collective: "
      f"{1e3 * t_drv:.3f} ms; same pose bits: {bool((T.as_array() == T2.as_array()).all())}; counters {drv.counters}")
