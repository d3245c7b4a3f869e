"""nn_stats.py for a 28k-point frame (four lanes per query): what the COLD first search of a call does against the warm
ones after it.  ICP_MI355X_LIB=icp_rust_amd/lib/libicp_mi355x_stats.so python3 profiles/nn_stats_frame.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import icp_rust_amd as I
from icp_rust_amd import synth
from icp_rust_amd.dist import HipStages, ShardedIcp
pk = synth.synthetic_scan3d_packets(150)
src, dst = synth.remove_invalid_values(pk[:75]), synth.remove_invalid_values(pk[75:150])
n = len(src)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
drv = ShardedIcp(HipStages(icp), n)
L = I.lib(); L.icp_debug_nn_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
out = (C.c_uint64 * 8)()
L.icp_debug_nn_hist.argtypes = [C.POINTER(C.c_uint64), C.c_int]
hist = (C.c_uint64 * 64)()
T = I.Transform(); drv.stages.prepare(d_src, T)
names = ["lanes", "row_bound_fetches", "record_batches", "exact_evals", "wave_loop_steps", "wave_cycles", "waves", "warm_queries"]
for it in range(6):
    L.icp_debug_nn_stats(out, 1); L.icp_debug_nn_hist(hist, 1)
    T, k = drv.step(d_src, T); torch.cuda.synchronize()
    L.icp_debug_nn_stats(out, 0)
    v = list(out); q = max(v[0], 1); w = max(v[6], 1)
    L.icp_debug_nn_hist(hist, 0); hv = list(hist)
    print(f"search {it}: per lane: rows {v[1]/q:.2f} batches {v[2]/q:.2f} exact {v[3]/q:.2f} | per wave: loop steps {v[4]/w:.1f} lifetime {v[5]/w:.0f} cycles; waves {v[6]}")
    print("   lanes by batches:", hv[:32]); print("   waves by max batches:", hv[32:])
