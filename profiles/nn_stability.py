"""How much of the correspondence set survives from one outer iteration to the next (benchmark pair and the
converging pair), and how far the queries move: the case for certifying an unchanged neighbour instead of
searching again.   python3 profiles/nn_stability.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import icp_rust_amd as I
from icp_rust_amd import synth

for name, (src, dst) in (("bench pair", synth.synthetic_pair(1_000_000, 1_000_000)),
                         ("converging pair", synth.converging_pair(1_000_000, 1_000_000)[:2])):
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    icp = I.Icp3d(d_dst)
    prev_T, prev_idx = None, None
    ext = np.abs(src[:, :2]).max()
    print(name, "extent", ext, "m", len(dst))
    for k in (1, 2, 3, 4, 5, 8, 12, 16, 20):
        T, idx, inner = icp.estimate(d_src, I.Transform(), k, return_info=True)
        A = T.as_array() if hasattr(T, "as_array") else np.asarray(T)
        if prev_T is not None:
            changed = float((idx != prev_idx).mean())
            cs = I.nn_cert_counters(icp)
            print(f"  iters {prev_k}->{k}: pose change {np.abs(A - prev_T).max():.3e}, indices changed {100 * changed:.3f} %, inner {inner[-1]}; "
                  f"certificates failed in the last search: {100 * cs[1] / len(src):.2f} % ({cs[0]} certified searches so far)")
        prev_T, prev_idx, prev_k = A, idx, k
    icp.close()
