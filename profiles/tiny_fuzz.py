"""Randomised parity run of the one-launch registration (k_tiny_estimate: n <= 1024 source points, m <= 2048 targets --
the reference's own 2-D scan sizes) and of its hand-overs to the host-driven path, against the oracle's tree variant,
bit for bit: sizes around every workgroup-size and capacity threshold, lattices (ties, runs of equal residuals),
far-off and large-rotation start poses, long inner loops, repeated calls on one handle.

    python3 profiles/tiny_fuzz.py [first_seed] [count]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import icp_rust_amd as I
import oracle_ffi as O

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = served = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(170_000 + seed)
    dim = 2 if seed % 3 else 3
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 511, 512, 513, 767, 768, 769, 1023, 1024, 1025, int(rng.integers(1, 1100))]))
    m = int(rng.choice([1, 2, 64, 2047, 2048, 2049, int(rng.integers(1, 2100))]))
    kind = int(rng.integers(0, 5))
    scale = np.array([10.0, 10.0, 1.0][:dim]) * rng.choice([1e-3, 1.0, 1e3])
    dst = rng.normal(size=(m, dim)) * scale
    if kind == 1:  # lattice: ties, duplicates, runs of equal residuals
        dst = np.round(rng.normal(size=(m, dim)) * 4) / 2 * scale
    elif kind == 2:  # collinear
        dst[:, 1] = 0.5 * dst[:, 0]
    src = dst[rng.integers(0, m, size=n)] + rng.normal(size=(n, dim)) * 0.02 * scale
    if kind == 3:  # exact copies: zero residuals, MAD = 0
        src = dst[rng.integers(0, m, size=n)].copy()
    p0 = rng.normal(size=3) * np.array([0.3, 0.3, 0.05]) * np.array([scale[0], scale[0], 1.0])
    if kind == 4:  # a large rotation to start from
        p0[2] = rng.uniform(-40.0, 40.0)
    T = I.Transform(p0)
    cls = I.Icp3d if dim == 3 else I.Icp2d
    icp = cls(np.ascontiguousarray(dst))
    for call in range(2):
        iters = int(rng.choice([0, 1, 2, 5, 20, 30]))
        try:
            Tn, idx, inner = icp.estimate(src, T, iters, return_info=True)
            got = ("ok", Tn.as_array(), idx, inner)
        except I._lib.IcpError as e:
            got = ("err", e.status)
        b, t = I.reduce_geometry(n)
        rc, oT, oidx, oinner = O.icp_estimate(dim, np.ascontiguousarray(dst), src, O.Pose(*T.pose.as_tuple()), iters,
                                              use_kdtree=False, sum_mode=1, reduce_blocks=b, reduce_threads=t)
        if rc == O.OK:
            same = got[0] == "ok" and np.array_equal(got[1], oT.as_array()) and np.array_equal(got[3], oinner[:len(got[3])]) \
                and (iters == 0 or np.array_equal(got[2], oidx))
        else:
            same = got[0] == "err" and got[1] == rc
        if not same:
            bad += 1
            print("TINY MISMATCH seed", seed, "dim", dim, "n", n, "m", m, "kind", kind, "call", call, "iters", iters, "gpu", got[0],
                  got[1] if got[0] == "err" else "", "oracle rc", rc)
        if got[0] == "ok":
            T = Tn
    served += icp.single_launch_counters()[0]
    icp.close()
print(f"tiny fuzz: seeds {first}..{first + count - 1}, {bad} mismatches, {served} calls served in one launch, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
