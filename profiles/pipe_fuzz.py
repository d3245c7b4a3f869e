"""Randomised consistency check of the PIPELINED sharded registration (csrc/pipe.hip, k_win_pick_shard): icp_create_multi with
1-8 virtual ranks against ONE handle on the same inputs -- pose, correspondence indices, inner counts bit for bit -- on
clouds of random size and shape, registrations long enough for the steady state (6-14 outer iterations), two calls per
object (the second starts pipelined from its first iteration), benchmark-shaped and converging pairs.
    python3 profiles/pipe_fuzz.py [first seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import icp_rust_amd as I
from icp_rust_amd import synth

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 50)
bad, piped, t0 = 0, 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(770_000 + seed)
    n = int(rng.choice([20_000, 66_000, 150_000, 300_001, int(rng.integers(30_000, 600_000))]))
    m = int(rng.choice([30_000, 120_000, int(rng.integers(20_000, 400_000))]))
    world = int(rng.choice([1, 2, 3, 4, 5, 8]))
    iters = int(rng.integers(6, 15))
    kind = "converging" if seed % 5 == 4 else "box"
    print("seed", seed, kind, "n", n, "m", m, "world", world, "iters", iters, file=sys.stderr, flush=True)
    if kind == "converging":
        src, dst = synth.converging_pair(n, m)[:2]
    else:
        src, dst = synth.synthetic_pair(n, m)
        if seed % 3 == 0:  # another truth motion, a rougher cloud
            src = src + rng.normal(size=src.shape) * 0.02
    init = I.Transform(rng.normal(size=3) * np.array([0.05, 0.05, 0.004])) if seed % 2 else I.Transform()
    one = I.Icp3d(dst)
    multi = I.IcpMulti(dst, [0] * world)
    ok = True
    for call in range(2):
        T1, idx1, in1 = one.estimate(src, init, iters, return_info=True)
        T2, idx2, in2 = multi.estimate(src, init, iters, return_info=True)
        ok = ok and np.array_equal(T1.as_array(), T2.as_array()) and np.array_equal(idx1, idx2) and np.array_equal(in1, in2)
    piped += multi.pipe_iterations()
    multi.close(); one.close()
    if not ok:
        bad += 1
        print("PIPE MISMATCH seed", seed, kind, "n", n, "m", m, "world", world, "iters", iters)
print(f"pipe fuzz: seeds {first}..{first + count - 1}, {bad} mismatches, {piped} outer iterations went through the pipeline, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
