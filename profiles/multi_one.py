"""icp_create_multi with W virtual ranks on the benchmark pair: per-call vs per-iteration cost
    python3 profiles/multi_one.py W"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import icp_rust_amd as I
from icp_rust_amd import synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1
src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
mu = I.IcpMulti(dst, [0] * W)
mu.estimate(src, I.Transform(), 5)
for k in (0, 1, 2, 5, 20, 20):
    t0 = time.perf_counter()
    mu.estimate(src, I.Transform(), k)
    print(f"W={W}: estimate({k}) {1e3 * (time.perf_counter() - t0):.3f} ms")
