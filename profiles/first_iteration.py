"""What the first outer iteration of an estimate call costs (cold search + an evaluation whose window
has to be predicted from the previous call): estimate(src, I, 1) and estimate(src, I, 20), repeated
on one handle, with the evaluation-path counters.  argv[1]: points (default 1M)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import icp_rust_amd as I
from icp_rust_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
src, dst = synth.synthetic_pair(n, n)
d_src = torch.from_numpy(src).cuda(); d_dst = torch.from_numpy(dst).cuda()
icp = I.Icp3d(d_dst)
def run(iters, reps):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        icp.estimate(d_src, I.Transform(), iters)
        ts.append(1e3 * (time.perf_counter() - t0))
    return ts
for iters in (20, 1, 20, 1, 2):
    c0 = I.gn_path_counters(icp)
    ts = run(iters, 4)
    c1 = I.gn_path_counters(icp)
    print(f"estimate(.., {iters:2d}) ms:", " ".join(f"{t:.3f}" for t in ts),
          " counters [windows tried, missed, short pipeline, radix, bets won, lost] +", [int(b - a) for a, b in zip(c0, c1)])
