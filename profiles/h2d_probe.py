import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import icp_rust_amd as I
from icp_rust_amd import synth
pk = synth.synthetic_scan3d_packets(150)
s3 = synth.remove_invalid_values(pk[:75]); d3 = synth.remove_invalid_values(pk[75:150])
icp = I.Icp3d(d3)
ds = torch.from_numpy(s3).cuda()
def med(f, reps=40):
    for _ in range(5): f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort(); return 1e3 * ts[len(ts) // 2]
print("host buffer   estimate(20): %.3f ms" % med(lambda: icp.estimate(s3, I.Transform(), 20)))
print("device tensor estimate(20): %.3f ms" % med(lambda: icp.estimate(ds, I.Transform(), 20)))
print("host buffer   estimate(1): %.3f ms" % med(lambda: icp.estimate(s3, I.Transform(), 1)))
print("device tensor estimate(1): %.3f ms" % med(lambda: icp.estimate(ds, I.Transform(), 1)))
print("host buffer   estimate(0): %.3f ms" % med(lambda: icp.estimate(s3, I.Transform(), 0)))
print("device tensor estimate(0): %.3f ms" % med(lambda: icp.estimate(ds, I.Transform(), 0)))
t = torch.empty_like(ds)
print("torch H2D copy of the cloud (pageable): %.3f ms" % med(lambda: (t.copy_(torch.from_numpy(s3)), torch.cuda.synchronize())))
