"""Extended randomised parity run (manual; the fixed-seed subset lives in tests/test_gpu_fuzz.py):
evaluation sequences over all pipelines and registrations across the size thresholds of the search
engines, everything bit-exact against the oracle's tree variant.  Prints one line per failure.

    python3 profiles/extended_fuzz.py [first_seed] [count]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import icp_rust_amd as I
import oracle_ffi as O
import test_gpu_fuzz as F
from parity_util import oracle_in_device_order  # (the sums of a call are folded in that call's fold order)

LOG = open(os.path.join(ROOT, "gpurun_out", "extended_fuzz_progress.log"), "w") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else sys.stderr


def note(*a):  # what is about to run: a device fault leaves its configuration behind
    print(*a, file=LOG, flush=True)


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
t0 = time.time()
# 1. single evaluations: sizes around every pipeline threshold (2, 1024, 4096, 65536, 4M is covered by tests)
for seed in range(first, first + count):
    rng = np.random.default_rng(50_000 + seed)
    n = int(rng.choice([2, 3, 7, 64, 65, 511, 512, 513, 1023, 1024, 1025, 2047, 4095, 4096, 4097, 9999, 65535, 65537,
                        int(rng.integers(2, 3000)), int(rng.integers(3000, 300_000))]))
    a = rng.normal(size=(n, 2)) * rng.uniform(1.0, 40.0)
    blocks, threads = I.reduce_geometry(n)
    kind = F.KINDS[seed % len(F.KINDS)]
    r = F.residuals(rng, n, kind)
    note("eval seed", seed, "n", n, kind)
    for step in range(4):
        r = r + rng.normal(size=2) * 1e-3 * (np.abs(r).mean() + 1e-6) if step != 2 else F.residuals(rng, n, F.KINDS[(seed + 1) % 6])
        T = I.Transform(rng.normal(size=3) * np.array([1e-3, 1e-3, 1e-5]))
        Tp = F.opose(T)
        Ta = np.stack([(Tp.r00 * a[:, 0] + Tp.r01 * a[:, 1]) + Tp.tx, (Tp.r10 * a[:, 0] + Tp.r11 * a[:, 1]) + Tp.ty], axis=1)
        b = Ta - r
        got = I.weighted_gauss_newton_update(T, a, b)
        rc, want, _ = O.weighted_gauss_newton_update_tree(Tp, a, b, blocks, threads)
        ok = (got is None and rc != O.OK) or (got is not None and rc == O.OK and np.array_equal(got, want))
        if not ok:
            bad += 1
            print("EVAL MISMATCH seed", seed, "n", n, "kind", kind, "step", step, got, want)
# 2. registrations across the search-engine thresholds
for seed in range(first, first + count):
    rng = np.random.default_rng(70_000 + seed)
    dim = 2 if seed % 4 == 0 else 3
    n = int(rng.choice([5, 100, 2047, 2048, 2049, 5000, 16383, 16384, 65535, 65536, 65537, int(rng.integers(50, 90_000))]))
    m = int(rng.choice([1, 2, 700, 2048, 2049, 8191, 8192, 8193, int(rng.integers(50, 60_000))]))
    dst = rng.normal(size=(m, dim)) * np.array([10.0, 10.0, 1.0][:dim])
    if seed % 5 == 0:
        dst = np.round(dst * 4) / 4  # lattice: ties and duplicates
    pick = rng.integers(0, m, size=n)
    src = dst[pick] + rng.normal(size=(n, dim)) * 0.05
    p = np.array([0.2, -0.1, 0.03]) * rng.uniform(0.2, 1.0)
    Tt = O.transform_new(p)
    src[:, :2] = O.transform_apply_many(O.transform_inverse(Tt), np.ascontiguousarray(src[:, :2]))
    iters = int(rng.integers(1, 7))
    note("registration seed", seed, "dim", dim, "n", n, "m", m, "iters", iters)
    icp = (I.Icp3d if dim == 3 else I.Icp2d)(dst)
    T, idx, inner = icp.estimate(src, I.Transform(), iters, return_info=True)
    rc, oT, oidx, oinner = oracle_in_device_order(icp, dim, dst, src, O.transform_identity(), iters)
    icp.close()
    if rc != O.OK or not (np.array_equal(idx, oidx) and np.array_equal(inner, oinner) and np.array_equal(T.as_array(), oT.as_array())):
        bad += 1
        print("REGISTRATION MISMATCH seed", seed, "dim", dim, "n", n, "m", m, "iters", iters, "rc", rc,
              "idx diff", int(np.sum(idx != oidx)) if rc == O.OK else None, inner.tolist(), None if rc != O.OK else oinner.tolist())
# 3. one long-lived handle: clouds of changing size, appends, engine switches, host and device buffers
import torch
for seed in range(first, first + max(count // 4, 1)):
    rng = np.random.default_rng(90_000 + seed)
    dim = 2 if seed % 3 == 0 else 3
    cls = I.Icp3d if dim == 3 else I.Icp2d
    scale = np.array([10.0, 10.0, 1.0][:dim])
    m = int(rng.choice([300, 2048, 2049, 8192, 9000, 30_000]))
    dst = rng.normal(size=(m, dim)) * scale
    on_device = bool(rng.integers(0, 2))
    keep = torch.from_numpy(dst).cuda() if on_device else None
    icp = cls(keep if on_device else dst)
    T = I.Transform([0.02, -0.01, 0.004])
    for step in range(5):
        act = int(rng.integers(0, 4))
        note("handle seed", seed, "dim", dim, "m", len(dst), "step", step, "act", act, "device", on_device)
        if act == 0:  # append (with or without a pose)
            k = int(rng.choice([1, 50, 3000, 9000]))
            extra = rng.normal(size=(k, dim)) * scale
            Ta = I.Transform(rng.normal(size=3) * 0.1) if rng.integers(0, 2) else None
            icp.append(torch.from_numpy(extra).cuda() if rng.integers(0, 2) else extra, Ta)
            if Ta is not None:
                r00, r10, r01, r11, tx, ty = Ta.pose.as_tuple()
                x, y = extra[:, 0].copy(), extra[:, 1].copy()
                extra[:, 0] = (r00 * x + r01 * y) + tx
                extra[:, 1] = (r10 * x + r11 * y) + ty
            dst = np.ascontiguousarray(np.concatenate([dst, extra]))
            continue
        if act == 1:  # engine switch
            mode = int(rng.choice([I.NN_AUTO, I.NN_BRUTE, I.NN_GRID]))
            I._lib.check(I.lib().icp_set_nn_mode(icp._h, mode), "icp_set_nn_mode")
            continue
        n = int(rng.choice([3, 900, 2048, 5000, 16384, 20_000, 65536, 70_000]))
        src = dst[rng.integers(0, len(dst), size=n)] + rng.normal(size=(n, dim)) * 0.05
        iters = int(rng.integers(0, 5))
        if act == 2:
            Tn, idx, inner = icp.estimate(src, T, iters, return_info=True)
        else:
            Tn, idx, inner = icp.estimate(torch.from_numpy(src).cuda(), T, iters, return_info=True)
        rc, oT, oidx, oinner = oracle_in_device_order(icp, dim, dst, src, O.Pose(*T.pose.as_tuple()), iters)
        same = rc == O.OK and np.array_equal(Tn.as_array(), oT.as_array()) and np.array_equal(inner, oinner[:len(inner)]) \
            and (iters == 0 or np.array_equal(idx, oidx))
        if not same:
            bad += 1
            print("HANDLE MISMATCH seed", seed, "dim", dim, "m", len(dst), "n", n, "iters", iters, "act", act, "rc", rc)
        T = Tn
    icp.close()
# 4. non-finite inputs: the reference panics on a NaN residual (src/stats.rs:12) -> a status, never a fault;
#    non-finite targets leave no grid -> the sweep serves, same indices as the oracle
for seed in range(first, first + max(count // 8, 1)):
    rng = np.random.default_rng(110_000 + seed)
    dim = 2 if seed % 2 else 3
    cls = I.Icp3d if dim == 3 else I.Icp2d
    m, n = int(rng.choice([500, 3000, 9000, 20_000])), int(rng.choice([100, 3000, 20_000, 70_000]))
    dst = rng.normal(size=(m, dim)) * 5
    src = dst[rng.integers(0, m, size=n)] + rng.normal(size=(n, dim)) * 0.05
    case = int(rng.integers(0, 3))
    note("nonfinite seed", seed, "dim", dim, "n", n, "m", m, "case", case)
    if case == 0:    # NaN in the source cloud
        src[rng.integers(0, n, size=3), rng.integers(0, dim)] = np.nan
    elif case == 1:  # +-inf targets
        dst[rng.integers(0, m, size=2), 0] = np.inf
        dst[rng.integers(0, m), 1] = -np.inf
    else:            # NaN target
        dst[rng.integers(0, m), dim - 1] = np.nan
    icp = cls(dst)
    try:
        Tn, idx, inner = icp.estimate(src, I.Transform(), 2, return_info=True)
        got = ("ok", Tn.as_array(), idx)
    except I._lib.IcpError as e:
        got = ("err", e.status, None)
    icp.close()
    b, t = I.reduce_geometry(n)
    rc, oT, oidx, _ = O.icp_estimate(dim, dst, src, O.transform_identity(), 2, use_kdtree=False, sum_mode=1, reduce_blocks=b,
                                     reduce_threads=t)
    if rc == O.OK:
        same = got[0] == "ok" and np.array_equal(got[1], oT.as_array()) and np.array_equal(got[2], oidx)
    else:
        same = got[0] == "err" and got[1] == rc
    if not same:
        bad += 1
        print("NONFINITE MISMATCH seed", seed, "dim", dim, "n", n, "m", m, "case", case, "gpu", got[0], got[1] if got[0] == "err" else "",
              "oracle rc", rc)
# 5. one handle, several calls beyond the one-lane-per-query threshold (seeded first search, warm searches, the
#    per-call window predictions), on clouds with structure and from poses that are far off
for seed in range(first, first + max(count // 16, 1)):
    rng = np.random.default_rng(130_000 + seed)
    dim = 2 if seed % 3 == 0 else 3
    cls = I.Icp3d if dim == 3 else I.Icp2d
    m = int(rng.choice([9000, 30_000, 70_000, 120_000]))
    shape = int(rng.integers(0, 4))
    if shape == 0:
        dst = rng.normal(size=(m, dim)) * np.array([10.0, 10.0, 1.0][:dim])
    elif shape == 1:  # lattice: ties and duplicates
        dst = np.round(rng.normal(size=(m, dim)) * 6) / 2
    elif shape == 2:  # two tight clusters and a sparse background
        dst = np.concatenate([rng.normal(size=(m // 2, dim)) * 0.05, rng.normal(size=(m // 4, dim)) * 0.05 + 30,
                              rng.uniform(-60, 60, size=(m - m // 2 - m // 4, dim))])
    else:  # flat in the last coordinate
        dst = rng.normal(size=(m, dim)) * 8
        dst[:, dim - 1] = 1.25
    dst = np.ascontiguousarray(dst)
    icp = cls(dst)
    T = I.Transform(rng.normal(size=3) * np.array([0.5, 0.5, 0.05]))
    for call in range(3):
        n = int(rng.choice([65_537, 70_000, 100_000]))
        src = dst[rng.integers(0, m, size=n)] + rng.normal(size=(n, dim)) * 0.03
        if call == 1:  # a different cloud in the same place: the previous call's predictions are of little use
            src = src + rng.normal(size=(n, dim)) * 0.5
        iters = int(rng.integers(1, 4))
        note("calls seed", seed, "dim", dim, "m", m, "shape", shape, "call", call, "n", n, "iters", iters)
        try:
            Tn, idx, inner = icp.estimate(src, T, iters, return_info=True)
            got = ("ok", Tn.as_array(), idx, inner)
        except I._lib.IcpError as e:
            got = ("err", e.status)
        if got[0] == "ok":
            rc, oT, oidx, oinner = oracle_in_device_order(icp, dim, dst, src, O.Pose(*T.pose.as_tuple()), iters)
        else:
            b, t = I.reduce_geometry(n)
            rc, oT, oidx, oinner = O.icp_estimate(dim, dst, src, O.Pose(*T.pose.as_tuple()), iters, use_kdtree=True, sum_mode=1,
                                                  reduce_blocks=b, reduce_threads=t)
        if rc == O.OK:
            same = got[0] == "ok" and np.array_equal(got[1], oT.as_array()) and np.array_equal(got[2], oidx) and \
                np.array_equal(got[3], oinner)
        else:
            same = got[0] == "err" and got[1] == rc
        if not same:
            bad += 1
            print("CALLS MISMATCH seed", seed, "dim", dim, "m", m, "shape", shape, "call", call, "n", n, "iters", iters,
                  "gpu", got[0], "oracle rc", rc)
        if got[0] == "ok":
            T = Tn
    icp.close()
print(f"extended fuzz: seeds {first}..{first + count - 1}, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
