import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/icp_rust_amd") else os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import icp_rust_amd as I
from icp_rust_amd.scans import load_scan2d
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
G = os.path.join(R, "tests", "golden", "scans2d")
src = load_scan2d(f"{G}/001.txt"); dst = load_scan2d(f"{G}/002.txt")
for _ in range(3):
    icp = I.Icp2d(dst); icp.estimate(src, I.Transform(), 20); icp.close()
def t(f, k=30):
    t0 = time.perf_counter()
    for _ in range(k): f()
    return 1e3 * (time.perf_counter() - t0) / k
def new_drop():
    icp = I.Icp2d(dst); icp.close()
def new_sync_drop():
    icp = I.Icp2d(dst); icp.synchronize(); icp.close()
def frame():
    icp = I.Icp2d(dst); icp.estimate(src, I.Transform(), 20); icp.close()
icp0 = I.Icp2d(dst)
print("new+drop", t(new_drop), "new+sync+drop", t(new_sync_drop), "frame (new+estimate20+drop)", t(frame), "estimate only", t(lambda: icp0.estimate(src, I.Transform(), 20)))
def nogc():
    return I.Icp2d(dst)
print("new without close (as bench_small)", t(nogc))
tn, ts, tc = [], [], []
for _ in range(30):
    t0 = time.perf_counter(); h = I.Icp2d(dst); t1 = time.perf_counter(); h.synchronize(); t2 = time.perf_counter(); h.close(); t3 = time.perf_counter()
    tn.append(t1 - t0); ts.append(t2 - t1); tc.append(t3 - t2)
med = lambda v: 1e3 * sorted(v)[len(v) // 2]
print("pieces: new", med(tn), "sync", med(ts), "close", med(tc))
import torch
tn, ts, tc = [], [], []
for _ in range(30):
    t0 = time.perf_counter(); h = I.Icp2d(dst); t1 = time.perf_counter(); h.close(); t3 = time.perf_counter()
    tn.append(t1 - t0); tc.append(t3 - t1)
print("pieces without sync: new", med(tn), "close", med(tc))
