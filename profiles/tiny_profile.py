import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import icp_rust_amd as I
from icp_rust_amd.scans import load_scan2d
G = "tests/golden/scans2d"
src = load_scan2d(f"{G}/001.txt"); dst = load_scan2d(f"{G}/002.txt")
icp = I.Icp2d(dst)
for _ in range(3):
    icp.estimate(src, I.Transform(), 20)
t0 = time.perf_counter()
for _ in range(10):
    icp.estimate(src, I.Transform(), 20)
print("ms per estimate(20):", 1e3 * (time.perf_counter() - t0) / 10)
