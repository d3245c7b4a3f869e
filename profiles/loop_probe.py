"""Timing probe for the one-launch inner loop (gn_loop.hip): the 1M benchmark pair, the converging pair and the
28k-point frame, `estimate(20)` each; with ICP_MI355X_LIB=icp_rust_amd/lib/libicp_ab_loopprof.so (built by
profiles/build_loop_variant.sh loopprof -DICP_LOOP_PROFILE) the kernel prints its phase stamps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import icp_rust_amd as I
from icp_rust_amd import synth

def timed(icp, src, iters=20, reps=8):
    for _ in range(2):
        icp.estimate(src, I.Transform(), iters)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        icp.estimate(src, I.Transform(), iters)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return 1e3 * ts[len(ts) // 2]

what = sys.argv[1:] or ["pair", "conv", "frame"]
if "pair" in what:
    src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    icp = I.Icp3d(d_dst)
    ms = timed(icp, d_src)
    print(f"pair1M: {ms / 20:.4f} ms/step  loop counters {I.gn_loop_counters(icp)} path {I.gn_path_counters(icp)}", flush=True)
    icp.close()
if "conv" in what:
    src, dst = synth.converging_pair(1_000_000, 1_000_000)[:2]
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    icp = I.Icp3d(d_dst)
    ms = timed(icp, d_src, reps=5)
    T, inner = icp.estimate(d_src, I.Transform(), 20, return_info="inner")
    print(f"converging: {ms / 20:.4f} ms/step inner {inner.tolist()} loop counters {I.gn_loop_counters(icp)} path {I.gn_path_counters(icp)}", flush=True)
    icp.close()
if "frame" in what:
    pk = synth.synthetic_scan3d_packets(150)
    s3 = synth.remove_invalid_values(pk[:75]); d3 = synth.remove_invalid_values(pk[75:150])
    icp = I.Icp3d(d3)
    ms = timed(icp, s3, reps=20)
    T, _, inner = icp.estimate(s3, I.Transform(), 20, return_info=True)
    print(f"frame28k: {ms:.3f} ms per estimate(20) inner {inner.tolist()} loop counters {I.gn_loop_counters(icp)} path {I.gn_path_counters(icp)}", flush=True)
    icp.close()
