"""How many waves of the LDS-tile search (nn_tile.hip) fall back to the gather walk, and what a search costs,
on the benchmark pair; run with ICP_GRID_OCC / ICP_GRID_FX to sweep the grid geometry.

    python3 profiles/tile_probe.py [iters]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import icp_rust_amd as I
from icp_rust_amd import synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
icp.estimate(d_src, I.Transform(), 3)
out = []
for k in (1, 2, 3, 5, 10, 20):
    icp.estimate(d_src, I.Transform(), k)
    out.append((k, I.nn_tile_counters(icp)))
print("fallback waves after k iterations:", out)
icp.profile_enable(1)
icp.profile_read()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    icp.estimate(d_src, I.Transform(), iters)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms, launches = icp.profile_read()
print(f"OCC={os.environ.get('ICP_GRID_OCC', '2')} FX={os.environ.get('ICP_GRID_FX', '4')} "
      f"TILE={os.environ.get('ICP_NN_TILE')} BLOCK={os.environ.get('ICP_QSORT_BLOCK')} NO_XCD={os.environ.get('ICP_TILE_NO_XCD')}: "
      f"{1e3 * dt / (reps * iters):.4f} ms/step, search {1e3 * ms / max(launches, 1):.1f} us avg over {launches} launches")
