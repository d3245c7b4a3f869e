"""Owners of the shared walk by the number of box rows their walk took (the -DICP_COOP_PROFILE variant:
    bash profiles/build_variant.sh coopprof nn_grid.hip -DICP_COOP_PROFILE; ICP_MI355X_LIB=.../libicp_ab_coopprof.so python3 profiles/coop_rows_hist.py)"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, numpy as np
import icp_rust_amd as I
from icp_rust_amd import synth
from icp_rust_amd.dist import HipStages, ShardedIcp
n = m = 1_000_000
src, dst = synth.synthetic_pair(n, m)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
drv = ShardedIcp(HipStages(icp), n)
L = C.CDLL(os.environ["ICP_MI355X_LIB"])
out = (C.c_uint64 * 32)()
prof = (C.c_uint64 * 12)()
T = I.Transform(); drv.stages.prepare(d_src, T)
for it in range(8):
    L.icp_debug_coop_rows(out, 1); L.icp_debug_coop_profile(prof, 1)
    T, k = drv.step(d_src, T); torch.cuda.synchronize()
    L.icp_debug_coop_rows(out, 0); L.icp_debug_coop_profile(prof, 0)
    pv = list(prof)
    print(f"   waves {pv[7]} rounds of rows {pv[8]} batches {pv[9]} flushes {pv[10]} candidates {pv[11]}")
    v = np.array(list(out), dtype=np.float64)
    tot = v.sum()
    if tot == 0: continue
    print(f"iter {it}: rows per owner %: " + " ".join(f"{i}:{100*v[i]/tot:.1f}" for i in range(0, 20) if v[i] > 0) + f"  >4: {100*v[5:].sum()/tot:.1f}%  >8: {100*v[9:].sum()/tot:.2f}%  mean {np.dot(v, np.arange(32))/tot:.2f}")
