import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import icp_rust_amd as I
from icp_rust_amd import synth
from icp_rust_amd.dist import HipStages, ShardedIcp, shard_range
n = m = 200_000
src, dst = synth.synthetic_pair(n, m)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
st = HipStages(icp)
# unsharded
drv = ShardedIcp(st, n)
T1, inner1 = drv.estimate(d_src, I.Transform(), 5)
# two shards emulated in one process
a = torch.empty((n, 2), dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
T = I.Transform(); inner2 = []
shards = [shard_range(n, r, 2) for r in range(2)]
srcs = [d_src[lo:hi].contiguous() for lo, hi in shards]
icps = [I.Icp3d(d_dst) for _ in shards]
for ic, s in zip(icps, srcs):
    ic.set_stream(torch.cuda.current_stream().cuda_stream); ic.prepare_source_device(s, T)
for it in range(5):
    for ic, s, (lo, hi) in zip(icps, srcs, shards):
        ic.correspond_device(s, T, a[lo:hi], b[lo:hi])
    dT, k = icp.estimate_transform_device(a, b)
    T = dT * T; inner2.append(k)
print("unsharded", T1.as_array(), inner1.tolist())
print("2 shards ", T.as_array(), inner2)
print("bit-identical:", np.array_equal(T1.as_array(), T.as_array()))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi as O
blocks, threads = I.reduce_geometry(n)
rc, oT, _, oin = O.icp_estimate(3, dst, src, O.transform_identity(), 5, use_kdtree=True, sum_mode=1, reduce_blocks=blocks, reduce_threads=threads)
print("oracle   ", oT.as_array(), oin.tolist())
T3, idx3, in3 = I.Icp3d(d_dst).estimate(d_src, I.Transform(), 5, return_info=True)
print("C++ loop ", T3.as_array(), in3.tolist())
