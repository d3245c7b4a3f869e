"""One-off scale check (not in the test-suite): 4M x 4M synthetic pair, GPU vs oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import icp_rust_amd as I
import oracle_ffi as O
from icp_rust_amd import synth
n = m = int(os.environ.get("N", 4_000_000))
src, dst = synth.synthetic_pair(n, m)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
t0 = time.perf_counter(); icp = I.Icp3d(d_dst); icp.synchronize(); print("Icp3d::new", time.perf_counter() - t0)
t0 = time.perf_counter(); T, idx, inner = icp.estimate(d_src, I.Transform(), 3, return_info=True); t_gpu = time.perf_counter() - t0
print("GPU estimate(3 it)", t_gpu, inner.tolist(), T.as_array())
t0 = time.perf_counter()
b, t = I.reduce_geometry(n)
rc, oT, oidx, oin = O.icp_estimate(3, dst, src, O.transform_identity(), 3, use_kdtree=True, sum_mode=1, reduce_blocks=b, reduce_threads=t)
print("oracle", time.perf_counter() - t0, oin.tolist(), oT.as_array())
print("idx equal:", np.array_equal(idx, oidx), " pose bit-equal:", np.array_equal(T.as_array(), oT.as_array()))
