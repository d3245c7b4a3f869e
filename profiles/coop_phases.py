"""Diagnostic: where a wave of the shared warm walk (k_nn_grid_warm_coop) spends its time on the benchmark pair.
Needs the variant  bash profiles/build_variant.sh coopprof nn_grid.hip "-DICP_COOP_DEFAULT=1 -DICP_COOP_PROFILE"
Run as:  ICP_MI355X_LIB=icp_rust_amd/lib/libicp_ab_coopprof.so python3 profiles/coop_phases.py"""
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
L = C.CDLL(os.environ["ICP_MI355X_LIB"]); L.icp_debug_coop_profile.argtypes = [C.POINTER(C.c_uint64), C.c_int]
out = (C.c_uint64 * 12)()
T = I.Transform(); drv.stages.prepare(d_src, T)
names = ["load+geometry", "row selection", "row bounds", "prefix+tables", "worker rounds", "flushes", "outputs"]
for it in range(8):
    L.icp_debug_coop_profile(out, 1)
    T, k = drv.step(d_src, T); torch.cuda.synchronize()
    L.icp_debug_coop_profile(out, 0)
    v = list(out); w = max(v[7], 1)
    if v[7] == 0: print(f"iter {it}: (not the shared walk)"); continue
    print(f"iter {it}: waves {v[7]} lifetime {sum(v[:7]) / w / 100:.2f} us = " + ", ".join(f"{nm} {v[j] / w / 100:.2f}" for j, nm in enumerate(names)) +
          f" | per wave: rounds of rows {v[8] / w:.2f}, worker rounds {v[9] / w:.2f}, flushes {v[10] / w:.2f}, candidates {v[11] / w:.1f}")
