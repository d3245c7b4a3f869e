"""One timing sample of the two regimes for whatever library ICP_MI355X_LIB names (profiles/ab_libs.py runs it
once per variant and round): the 28k-point frame (estimate(20), host arrays in) and the 1M pair (device arrays)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import icp_rust_amd as I
from icp_rust_amd import synth

what = sys.argv[1] if len(sys.argv) > 1 else "both"
out = []
if what in ("both", "frame"):
    pk = synth.synthetic_scan3d_packets(150)
    s3 = synth.remove_invalid_values(pk[:75]); d3 = synth.remove_invalid_values(pk[75:150])
    icp = I.Icp3d(d3)
    for _ in range(5):
        icp.estimate(s3, I.Transform(), 20)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter()
        icp.estimate(s3, I.Transform(), 20)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    out.append(f"frame28k min {1e3 * ts[0]:.3f} med {1e3 * ts[len(ts) // 2]:.3f} ms")
    sh = np.ascontiguousarray(s3[np.random.default_rng(5).permutation(len(s3))])  # the same frame in random order
    for _ in range(3):
        icp.estimate(sh, I.Transform(), 20)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        icp.estimate(sh, I.Transform(), 20)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    out.append(f"shuffled med {1e3 * ts[len(ts) // 2]:.3f} ms")
    icp.close()
if what in ("both", "pair"):
    src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    icp = I.Icp3d(d_dst)
    for _ in range(2):
        icp.estimate(d_src, I.Transform(), 20)
    ts = []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        icp.estimate(d_src, I.Transform(), 20)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20)
    ts.sort()
    t40 = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        icp.estimate(d_src, I.Transform(), 40)
        torch.cuda.synchronize()
        t40.append(time.perf_counter() - t0)
    t40.sort()
    slope = (t40[len(t40) // 2] - 20 * ts[len(ts) // 2]) / 20
    out.append(f"pair1M min {1e3 * ts[0]:.4f} med {1e3 * ts[len(ts) // 2]:.4f} ms/step; marginal step {1e3 * slope:.4f} ms, "
               f"per-call {1e3 * (20 * ts[len(ts) // 2] - 20 * slope):.3f} ms")
if what in ("both", "conv"):
    src, dst, _ = synth.converging_pair(1_000_000, 1_000_000)
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    icp = I.Icp3d(d_dst)
    icp.estimate(d_src, I.Transform(), 20)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T, idx, inner = icp.estimate(d_src, I.Transform(), 20, return_info=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    ev = int(np.sum(inner)) + 20
    out.append(f"converging med {1e3 * ts[len(ts) // 2]:.2f} ms/call, {ev} evaluations -> {1e6 * ts[len(ts) // 2] / ev:.1f} us each")
print(" | ".join(out), flush=True)
