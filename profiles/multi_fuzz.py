"""Randomised check of the sharded evaluation: icp_create_multi with virtual ranks on one GPU and the
block-sharded Python driver (dist.BlockShardedIcp, LocalComm) against ONE handle on the same inputs -- pose,
indices and inner counts must be identical to the bit, whatever the number of ranks.

    python3 profiles/multi_fuzz.py [first_seed] [count]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import icp_rust_amd as I
from icp_rust_amd.dist import BlockShardedIcp, HipStages, LocalComm, local_indices

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(150_000 + seed)
    dim = 2 if seed % 4 == 0 else 3
    n = int(rng.choice([3, 600, 5000, 40_000, 70_000, 300_000, int(rng.integers(2000, 200_000))]))
    m = int(rng.choice([50, 3000, 9000, 60_000, int(rng.integers(100, 100_000))]))
    world = int(rng.choice([1, 2, 3, 5, 8, 12, 16]))
    dst = rng.normal(size=(m, dim)) * np.array([10.0, 10.0, 1.0][:dim])
    if seed % 5 == 0:
        dst = np.round(dst * 2) / 2
    src = dst[rng.integers(0, m, size=n)] + rng.normal(size=(n, dim)) * 0.05
    T0 = I.Transform(rng.normal(size=3) * np.array([0.3, 0.3, 0.03]))
    iters = int(rng.integers(1, 5))
    cls = I.Icp3d if dim == 3 else I.Icp2d
    print("seed", seed, "dim", dim, "n", n, "m", m, "world", world, "iters", iters, file=sys.stderr, flush=True)
    one = cls(dst)
    T1, idx1, in1 = one.estimate(src, T0, iters, return_info=True)
    one.close()
    multi = I.IcpMulti(dst, [0] * world, dim=dim)
    T2, idx2, in2 = multi.estimate(src, T0, iters, return_info=True)
    multi.close()
    ok_multi = np.array_equal(T1.as_array(), T2.as_array()) and np.array_equal(idx1, idx2) and np.array_equal(in1, in2)
    d_dst = torch.from_numpy(dst).cuda()
    handles = {r: cls(d_dst) for r in range(world)}
    drv = BlockShardedIcp({r: HipStages(h) for r, h in handles.items()}, n, world, LocalComm(world))
    T3, in3, perms = drv.estimate_full(torch.from_numpy(src).cuda(), T0, iters)  # (fold order of the call, then shards)
    idx3s = np.zeros(max(n, 1), dtype=np.uint32)
    for r, li in drv.last_indices().items():
        idx3s[local_indices(n, r, world)] = li.cpu().numpy().view(np.uint32)
    idx3 = np.zeros(max(n, 1), dtype=np.uint32)
    if n:
        idx3[perms[0].cpu().numpy().astype(np.int64)] = idx3s[:n]
    for h in handles.values():
        h.close()
    ok_drv = np.array_equal(T1.as_array(), T3.as_array()) and np.array_equal(np.asarray(in1), np.asarray(in3)) and \
        np.array_equal(idx1, idx3[:n])
    if not (ok_multi and ok_drv):
        bad += 1
        print("MULTI MISMATCH seed", seed, "dim", dim, "n", n, "m", m, "world", world, "iters", iters, "multi ok", ok_multi,
              "driver ok", ok_drv)
print(f"multi fuzz: seeds {first}..{first + count - 1}, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
