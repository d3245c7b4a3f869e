"""Where a frame of the scan3d loop (examples/scan3d.rs:104-158 = harness.run_scan3d, serial) spends its time on the host's
clock: packets -> filtered cloud, Icp3d::new, estimate(src, T, 20), drop, inverse.  usage: python3 profiles/frame_loop_parts.py"""
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
import icp_rust_amd as I
from icp_rust_amd import synth
from icp_rust_amd.harness import remove_invalid_values

pk = synth.synthetic_scan3d_packets(75 * 14)
src = remove_invalid_values(pk[:75])
T = I.Transform.identity()
rows = []
for k in range(13):
    t0 = time.perf_counter()
    dst = remove_invalid_values(pk[75 * k:75 * (k + 1)])
    t1 = time.perf_counter()
    icp = I.Icp3d(dst)
    t2 = time.perf_counter()
    T, inner = icp.estimate(src, T, 20, return_info="inner")
    t3 = time.perf_counter()
    icp.close()
    t4 = time.perf_counter()
    inv = T.inverse()
    t5 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, list(inner)))
print("frame: filter, Icp3d::new, estimate(20), drop, inverse [ms]; inner counts")
for k, r in enumerate(rows):
    print(k, " ".join(f"{1e3 * x:6.3f}" for x in r[:5]), r[5])
m = np.median(np.array([r[:5] for r in rows[2:]]), axis=0)
print("median of frames 2..:", " ".join(f"{1e3 * x:6.3f}" for x in m), "sum", f"{1e3 * m.sum():.3f}")
