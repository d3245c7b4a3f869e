"""What a NEW handle per frame costs on top of the registration (examples/scan3d.rs creates an Icp3d per frame): per-frame
time with a fresh handle (from the pool) against the same registration on a handle that has seen the previous call,
and which evaluation pipelines served the fresh handle's call (icp_gn_path_counters).
    python3 profiles/frame_loop_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_rust_amd as I
from icp_rust_amd import synth

pk = synth.synthetic_scan3d_packets(75 * 14)
frames = [synth.remove_invalid_values(pk[75 * k:75 * (k + 1)]) for k in range(14)]
for _ in range(2):  # warm the pool
    icp = I.Icp3d(frames[0]); icp.estimate(frames[1], I.Transform(), 20); icp.close()
t_new, t_est, t_close, ctr = [], [], [], []
for k in range(1, 13):
    t0 = time.perf_counter(); icp = I.Icp3d(frames[k]); t1 = time.perf_counter()
    icp.estimate(frames[k + 1], I.Transform(), 20); t2 = time.perf_counter()
    ctr.append(I.gn_path_counters(icp)); icp.close(); t3 = time.perf_counter()
    t_new.append(t1 - t0); t_est.append(t2 - t1); t_close.append(t3 - t2)
icp = I.Icp3d(frames[5])
for _ in range(3):
    icp.estimate(frames[6], I.Transform(), 20)
t0 = time.perf_counter()
for _ in range(10):
    icp.estimate(frames[6], I.Transform(), 20)
t_warm = (time.perf_counter() - t0) / 10
med = lambda v: sorted(v)[len(v) // 2]
print(f"fresh handle per frame: new {1e3 * med(t_new):.3f} ms + estimate(20) {1e3 * med(t_est):.3f} ms + drop {1e3 * med(t_close):.3f} ms; "
      f"the same estimate on a handle that has run the call before: {1e3 * t_warm:.3f} ms; "
      f"pipelines of a fresh handle's call (window tried, missed, pull, radix, ...): {ctr[-1]}")
