"""Diagnostic: what the WARM grid search does per launch on the benchmark pair (needs `make -C icp_rust_amd/csrc stats`).
Run as:  ICP_MI355X_LIB=icp_rust_amd/lib/libicp_mi355x_stats.so python profiles/nn_warm_stats.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import icp_rust_amd as I
from icp_rust_amd import synth
from icp_rust_amd.dist import HipStages, ShardedIcp
n = m = 1_000_000
src, dst = synth.synthetic_pair(n, m)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
drv = ShardedIcp(HipStages(icp), n)
L = I.lib(); L.icp_debug_nn_warm.argtypes = [C.POINTER(C.c_uint64), C.c_int]
out = (C.c_uint64 * 72)()
T = I.Transform(); drv.stages.prepare(d_src, T)
for it in range(12):
    L.icp_debug_nn_warm(out, 1)
    T, k = drv.step(d_src, T); torch.cuda.synchronize()
    L.icp_debug_nn_warm(out, 0)
    v = list(out); w = max(v[7], 1); q = n
    print(f"iter {it}: waves {v[7]}  per wave: row groups {v[0]/w:.2f} chunks {v[1]/w:.2f} (with an exact test {v[2]/w:.2f}) "
          f"lifetime {v[6]/w:.0f} ticks | per lane: row groups {v[3]/q:.2f} chunks {v[4]/q:.2f} exact tests {v[5]/q:.3f}")
    if it in (1, 5, 11):
        print("   waves by row groups:", v[8:40]); print("   waves by chunks:    ", v[40:72])
