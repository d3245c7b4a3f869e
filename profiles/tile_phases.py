"""In-kernel phase profile of the LDS-tile search (diagnostic build: make -C icp_rust_amd/csrc stats).
    ICP_MI355X_LIB=icp_rust_amd/lib/libicp_mi355x_stats.so python3 profiles/tile_phases.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import icp_rust_amd as I
from icp_rust_amd import synth
src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
icp.estimate(d_src, I.Transform(), 10)
torch.cuda.synchronize()
L = I.lib()
buf = np.zeros((16384, 8), dtype=np.uint64)
L.icp_debug_tile_profile.argtypes = [C.c_void_p]
assert L.icp_debug_tile_profile(buf.ctypes.data) == 0
nw = 15625
p = buf[:nw].astype(np.int64)
done = p[:, 7] != 0
print("waves", nw, "completed in the tile kernel", int(done.sum()))
names = ["load q/prev + box", "unions", "table staged", "records staged", "items built", "items screened", "exact + stores"]
d = np.diff(p[done], axis=1)
for i, nm in enumerate(names):
    print(f"  {nm:22s} mean {d[:, i].mean():9.0f}  median {np.median(d[:, i]):9.0f}  p95 {np.percentile(d[:, i], 95):9.0f} cycles")
life = p[done, 7] - p[done, 0]
print(f"  lifetime               mean {life.mean():9.0f}  median {np.median(life):9.0f}  p95 {np.percentile(life, 95):9.0f}")
span = p[done, 7].max() - p[:, 0].min()
print("kernel span (cycles, stamps of all XCDs are one clock?)", int(span))
gv = p[~done]
print("gave up:", len(gv), "mean cycles to the last stamp they wrote", float((gv[:, 1:3].max(axis=1) - gv[:, 0]).mean()))
