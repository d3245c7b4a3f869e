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

# how many waves are resident over the launch: the (start, end) of every workgroup of the last search
import numpy as np
nb = min((n + 63) // 64, 16384)
L.icp_debug_coop_spans.argtypes = [C.c_void_p, C.c_int]
sp = np.zeros((nb, 2), dtype=np.int64)
L.icp_debug_coop_spans(sp.ctypes.data, nb)
t0 = sp[:, 0].min(); sp -= t0
if os.environ.get('COOP_SPANS_OUT'): np.save(os.environ['COOP_SPANS_OUT'], sp)
dur = sp[:, 1].max()
print(f"last launch: {nb} waves, first start to last end {dur / 100:.1f} us; wave lifetime mean {np.mean(sp[:, 1] - sp[:, 0]) / 100:.2f} us, "
      f"p50 {np.median(sp[:, 1] - sp[:, 0]) / 100:.2f}, p90 {np.percentile(sp[:, 1] - sp[:, 0], 90) / 100:.2f}, max {np.max(sp[:, 1] - sp[:, 0]) / 100:.2f}")
edges = np.linspace(0, dur, 21)
for a_, b_ in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a_ + b_)
    res = int(np.sum((sp[:, 0] <= mid) & (sp[:, 1] > mid)))
    started = int(np.sum((sp[:, 0] >= a_) & (sp[:, 0] < b_)))
    print(f"  t = {mid / 100:6.1f} us: resident waves {res:5d} ({res / 1024:.2f} per SIMD), started in the interval {started}")
