"""Timing of the reference-sized workloads (BASELINE configs[0] scale: a 2-D scan pair; configs[1]: a
28.8k-point 3-D window): GPU estimate(20 iterations) next to the single-thread CPU oracle.
Not the headline benchmark (bench.py); a tool to keep the small-problem latency honest."""
import sys, time, os
ROOT = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import icp_rust_amd as I
import oracle_ffi as O
from icp_rust_amd import synth, harness
from icp_rust_amd.scans import load_scan2d
G = os.path.join(ROOT, 'tests', 'golden', 'scans2d')
src = load_scan2d(f'{G}/001.txt'); dst = load_scan2d(f'{G}/002.txt')
icp = I.Icp2d(dst)
T = icp.estimate(src, I.Transform(), 20)
t0 = time.perf_counter()
for _ in range(20): T, idx, inner = icp.estimate(src, I.Transform(), 20, return_info=True)
t = (time.perf_counter() - t0) / 20
print(f"2D scan ({len(src)}x{len(dst)}): GPU estimate(20 it) {t*1e3:.3f} ms, inner {inner.tolist()}")
for _ in range(2):  # (the first handle of a process allocates its buffers, streams and pinned memory: ~11 ms, once)
    tmp = I.Icp2d(dst); tmp.estimate(src, I.Transform(), 20); tmp.close()
t0 = time.perf_counter()
for _ in range(20):
    tmp = I.Icp2d(dst); tmp.estimate(src, I.Transform(), 20); tmp.close()
tf = (time.perf_counter() - t0) / 20
# (new + drop timed where examples/scan2d.rs pays them: around an estimate, handles recycled through the pool)
print(f"   Icp2d::new + estimate(20) + drop {tf*1e3:.3f} ms -> new + drop {1e3*(tf-t):.3f} ms")
tree = O.KdTree(dst)
t0 = time.perf_counter()
for _ in range(20): rc, oT, _, oin = tree.estimate(src, O.transform_identity(), 20)
print(f"   CPU oracle estimate(20 it) {1e3*(time.perf_counter()-t0)/20:.3f} ms, inner {oin.tolist()}")
pk = synth.synthetic_scan3d_packets(150)
s3 = synth.remove_invalid_values(pk[:75]); d3 = synth.remove_invalid_values(pk[75:150])
icp3 = I.Icp3d(d3)
icp3.estimate(s3, I.Transform(), 20)
t0 = time.perf_counter()
for _ in range(10): T3, idx, inner = icp3.estimate(s3, I.Transform(), 20, return_info=True)
print(f"3D scan ({len(s3)}x{len(d3)}): GPU estimate(20 it) {1e3*(time.perf_counter()-t0)/10:.3f} ms, inner {inner.tolist()}")
t0 = time.perf_counter()
tf3 = 0.
for k in range(12):
    if k == 2: t0 = time.perf_counter()  # (two untimed frames: first-use allocations)
    tmp = I.Icp3d(d3); tmp.estimate(s3, I.Transform(), 20); tmp.close()
tf3 = (time.perf_counter() - t0) / 10
print(f"   Icp3d::new + estimate(20) + drop {tf3*1e3:.3f} ms (a fresh handle per frame, as examples/scan3d.rs has it)")
t0=time.perf_counter(); tree3 = O.KdTree(d3); tb=time.perf_counter()-t0
t0 = time.perf_counter()
for _ in range(3): rc, oT, _, oin = tree3.estimate(s3, O.transform_identity(), 20)
print(f"   CPU oracle estimate(20 it) {1e3*(time.perf_counter()-t0)/3:.3f} ms (+ kd build {tb*1e3:.2f} ms), inner {oin.tolist()}")
# the reference's frame loop itself (examples/scan3d.rs:104-158): a new Icp3d per frame, warm-started
# estimate(src, T, 20); handle turnover comes out of the pool after the first frame
pk = synth.synthetic_scan3d_packets(75 * 12)
harness.run_scan3d(pk[:75 * 3])
for piped in (False, True):
    tm = []
    t0 = time.perf_counter()
    Ts, inv, path = harness.run_scan3d(pk, pipeline=piped, timings=tm)
    nf = len(Ts)
    print(f"scan3d frame loop ({nf} frames of ~28k points, {'frame k+1 created while frame k estimates' if piped else 'serial'}): "
          f"{1e3*(time.perf_counter()-t0)/nf:.3f} ms per frame (Icp3d::new + estimate(20) + drop); "
          f"per-frame ms {[round(1e3*x, 2) for x in tm]}")
    if not piped:
        ref = [t.as_array() for t in Ts]
    else:
        assert all(np.array_equal(a, t.as_array()) for a, t in zip(ref, Ts)), "pipelining changed the trajectory"
# the same stream from a packet container on disk (scans.hdf5 stand-in), pipelined
import tempfile
from icp_rust_amd import scans
with tempfile.TemporaryDirectory() as d:
    f = os.path.join(d, "scans.icppkt")
    scans.write_packets(f, pk)
    t0 = time.perf_counter()
    Tf, _, _ = harness.run_scan3d(scans.PacketFile(f))
    print(f"scan3d frame loop from a packet file: {1e3*(time.perf_counter()-t0)/len(Tf):.3f} ms per frame")
    assert all(np.array_equal(a, t.as_array()) for a, t in zip(ref, Tf))
