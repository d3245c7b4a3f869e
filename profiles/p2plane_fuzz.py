"""Randomised check of the point-to-plane extension on a map that grows (icp_compute_target_normals,
icp_append_targets, icp_update_target_normals, icp_estimate_point_to_plane) against its incremental CPU statement
(oracle: orc_p2pl_normals_range / orc_p2pl_estimate).  Not a parity claim -- the reference has no normals -- but the
device and the CPU statement of the SAME definition must agree: indices and inner counts exactly, normals and poses
to rounding (tree sums vs left folds).

    python3 profiles/p2plane_fuzz.py [first_seed] [count]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import icp_rust_amd as I
import oracle_ffi as O


def room(rng, m, noise=2e-3):
    """points on the walls and the floor of a box, with sensor noise"""
    side = rng.integers(0, 5, size=m)
    u, v = rng.uniform(-4, 4, size=m), rng.uniform(0, 2.5, size=m)
    p = np.zeros((m, 3))
    p[side == 0] = np.stack([u, np.full(m, 4.0), v], 1)[side == 0]
    p[side == 1] = np.stack([u, np.full(m, -4.0), v], 1)[side == 1]
    p[side == 2] = np.stack([np.full(m, 4.0), u, v], 1)[side == 2]
    p[side == 3] = np.stack([np.full(m, -4.0), u, v], 1)[side == 3]
    p[side == 4] = np.stack([u, rng.uniform(-4, 4, size=m), np.zeros(m)], 1)[side == 4]
    return p + rng.normal(size=(m, 3)) * noise


def moved(p, T):
    q = np.array(p, dtype=np.float64)
    r = T.pose
    x, y = q[:, 0].copy(), q[:, 1].copy()
    q[:, 0] = (r.r00 * x + r.r01 * y) + r.tx
    q[:, 1] = (r.r10 * x + r.r11 * y) + r.ty
    return q


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
t0 = time.time()
O.set_threads(16)
for seed in range(first, first + count):
    rng = np.random.default_rng(190_000 + seed)
    m0 = int(rng.choice([40, 700, 2500, 6000]))
    k = int(rng.integers(3, 17))
    dst = room(rng, m0)
    icp = I.Icp3d(dst)
    icp.compute_normals(k)
    normals = O.p2pl_normals(dst, k)
    T = I.Transform()
    for frame in range(int(rng.integers(1, 4))):
        n = int(rng.choice([30, 500, 3000]))
        scan = moved(room(rng, n), I.Transform(rng.normal(size=3) * np.array([0.03, 0.03, 0.01])).inverse())
        iters = int(rng.integers(1, 6))
        Tn, idx, inner = icp.estimate_point_to_plane(scan, T, iters, return_info=True)
        rc, oT, oidx, oinner = O.p2pl_estimate(O.KdTree(dst), normals, scan, O.Pose(*T.pose.as_tuple()), iters)
        ok = rc == O.OK and np.array_equal(idx, oidx) and np.array_equal(inner, oinner) and \
            np.max(np.abs(Tn.as_array() - oT.as_array())) < 1e-8
        extra = moved(scan, Tn)
        icp.append(scan, Tn)
        dst = np.ascontiguousarray(np.concatenate([dst, extra]))
        if rng.integers(0, 4) == 0:  # now and then all normals again, from the current cloud
            icp.compute_normals(k)
            normals = O.p2pl_normals(dst, k)
        else:
            icp.update_normals(k)
            normals = O.p2pl_normals_update(dst, len(normals), k, normals)
        got = icp.read_normals()
        ok = ok and np.max(np.abs(got - normals)) < 1e-6 and np.max(np.abs(icp.read_targets() - dst)) < 1e-9
        if not ok:
            bad += 1
            print("P2PLANE MISMATCH seed", seed, "m0", m0, "k", k, "frame", frame, "n", n, "iters", iters, "rc", rc,
                  "idx diff", int(np.sum(idx != oidx)) if rc == O.OK else None,
                  "pose diff", float(np.max(np.abs(Tn.as_array() - oT.as_array()))) if rc == O.OK else None,
                  "normals diff", float(np.max(np.abs(got - normals))))
            break
        normals = got  # (carry the device's values: the comparison is per step, not of accumulated rounding)
        dst = icp.read_targets()
        T = Tn
    icp.close()
O.set_threads(1)
print(f"p2plane fuzz: seeds {first}..{first + count - 1}, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
