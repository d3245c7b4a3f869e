import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import icp_rust_amd as I
import oracle_ffi as O
from icp_rust_amd import synth
from icp_rust_amd.dist import shard_range
n = m = 200_000
src, dst = synth.synthetic_pair(n, m)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
cs = torch.cuda.current_stream().cuda_stream
full = I.Icp3d(d_dst); full.set_stream(cs)
shards = [shard_range(n, r, 2) for r in range(2)]
srcs = [d_src[lo:hi].contiguous() for lo, hi in shards]
icps = [I.Icp3d(d_dst) for _ in shards]
for ic in icps: ic.set_stream(cs)
T = I.Transform()
full.prepare_source_device(d_src, T)
for ic, s in zip(icps, srcs): ic.prepare_source_device(s, T)
tree = O.KdTree(dst)
for it in range(3):
    a1 = torch.zeros((n, 2), dtype=torch.float64, device="cuda"); b1 = torch.zeros_like(a1); i1 = torch.zeros(n, dtype=torch.int32, device="cuda")
    a2 = torch.zeros_like(a1); b2 = torch.zeros_like(a1); i2 = torch.zeros_like(i1)
    full.correspond_device(d_src, T, a1, b1, i1)
    for ic, s, (lo, hi) in zip(icps, srcs, shards):
        ic.correspond_device(s, T, a2[lo:hi], b2[lo:hi], i2[lo:hi])
    torch.cuda.synchronize()
    p = T.pose
    st = src.copy(); st[:, 0] = (p.r00 * src[:, 0] + p.r01 * src[:, 1]) + p.tx; st[:, 1] = (p.r10 * src[:, 0] + p.r11 * src[:, 1]) + p.ty
    rc, oi = tree.search(st)
    print(it, "full==oracle idx", np.array_equal(i1.cpu().numpy().view(np.uint32), oi), "shard==oracle idx", np.array_equal(i2.cpu().numpy().view(np.uint32), oi),
          "a eq", torch.equal(a1, a2), "b eq", torch.equal(b1, b2))
    dT, k = full.estimate_transform_device(a1, b1)
    dT2, k2 = full.estimate_transform_device(a2, b2)
    print("   dT equal", np.array_equal(dT.as_array(), dT2.as_array()), k, k2)
    T = dT * T
