#!/usr/bin/env python3
"""Benchmark of the ICP hot path on MI355X: outer ICP iterations per second on the
synthetic 1M-vs-1M 3-D pair of BASELINE.json (configs[2]; sharded over N GPUs = configs[3]).

A "step" is one outer iteration of Icp3d::estimate (src/lib.rs:155-171): transform the
source cloud, exact nearest-neighbour match against the target cloud, run the whole inner
Huber/MAD Gauss-Newton loop, compose the pose.  Inputs are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 2
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 (public spec)
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X vector FP32 (MI355X_MICROARCH.md chip table)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
MAX_ITER = 20                  # outer iterations per estimate() call (both reference examples pass 20)


def parity_vs_oracle(tree, src, gpu, n_iter, blocks, threads, perm=None):
    """Part of the CPU-baseline leg, outside every timed region: the registration the GPU just ran
    (`n_iter` outer iterations from the identity pose), repeated by the oracle in the device's
    documented summation order.  Pose, correspondence indices and inner-iteration counts must be
    the same bits."""
    import oracle_ffi as O

    cores = os.cpu_count() or 1
    O.set_threads(cores)
    # the device folds its sums over the source cloud in fold order (icp_last_fold_order: the cell-sorted
    # snapshot of the call); the oracle folds over the order of the cloud it is handed
    if perm is None:
        perm = np.arange(len(src))
    try:
        rc, oT, oidx_s, oinner = tree.estimate(np.ascontiguousarray(src[perm]), O.transform_identity(), n_iter,
                                               O.IcpOpts(1, 1, blocks, threads))
    finally:
        O.set_threads(1)
    oidx = np.empty_like(oidx_s)
    oidx[perm] = oidx_s
    T, idx, inner = gpu
    return {"checked": "estimate(src, identity, %d) on the benchmark pair vs the oracle in device summation order "
                       "(%d x %d)" % (n_iter, blocks, threads),
            "oracle_rc": int(rc),
            "pose_bits_equal": bool(np.array_equal(T.as_array(), oT.as_array())),
            "idx_equal": bool(np.array_equal(idx, oidx)),
            "inner_iterations_equal": bool(np.array_equal(np.asarray(inner, dtype=np.uint32), oinner)),
            "pose_max_abs_diff": float(np.max(np.abs(T.as_array() - oT.as_array())))}


def cpu_baseline(src, dst, iters, parity_of=None):
    """The oracle (kd-tree exact NN + the reference-order inner loop), single thread, timed on
    this host: `iters` outer iterations of the same workload from the identity pose.  This is
    the only place bench.py touches oracle/ -- as the reported baseline, never as the thing
    measured.  `all_cores` repeats it with the queries of each search split over every host core
    (the reference itself is single-threaded; SURVEY.md 8(d) asks for both)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O

    t0 = time.perf_counter()
    tree = O.KdTree(dst)
    t_build = time.perf_counter() - t0

    def run(threads):
        O.set_threads(threads)
        tree.search(src[:4096])  # thread start-up is not part of the sample
        T = O.transform_identity()
        times = []
        for _ in range(iters):
            t0 = time.perf_counter()
            rc, T, _, _ = tree.estimate(src, T, 1)
            times.append(time.perf_counter() - t0)
            assert rc == O.OK
        O.set_threads(1)
        return float(np.mean(times)), times

    per_iter, times = run(1)
    cores = os.cpu_count() or 1
    per_iter_all = run(cores)[0] if cores > 1 else per_iter
    parity = parity_vs_oracle(tree, src, *parity_of) if parity_of is not None else None
    return {
        "parity": parity,
        "value": 1.0 / per_iter,
        "unit": "iterations/s",
        "cores": 1,
        "kind": "port",
        "all_cores": {"value": 1.0 / per_iter_all, "cores": cores,
                      "note": "kd-tree queries split over all host cores, inner Gauss-Newton loop serial"},
        "sample": f"{iters} outer iterations of the same {len(src)}x{len(dst)} pair from the identity pose "
                  f"(oracle restatement, not the Rust binary; kd-tree build {t_build:.2f} s not included)",
        "kdtree_build_s": t_build,
        "s_per_iteration": times,
    }


def gn_large(npts):
    """the Gauss-Newton reduce kernels alone on a pair list far larger than the
    256 MiB Infinity Cache, where HBM bandwidth -- not launch latency -- is what they run against.
    One estimate_transform call on `npts` device-generated pairs; bytes per evaluation as SURVEY
    prices them (96 B/point: 32 in + 16 out for the residual pass, 48 in for the accumulate pass;
    the selection passes in between are implementation-defined and not counted)."""
    import torch
    import icp_rust_amd as I

    g = torch.Generator(device="cuda").manual_seed(1234)
    a = (torch.rand((npts, 2), dtype=torch.float64, device="cuda", generator=g) - 0.5) * 80.0
    c, s_ = np.cos(0.015), np.sin(0.015)
    b = torch.empty_like(a)
    b[:, 0] = c * a[:, 0] - s_ * a[:, 1] + 0.3
    b[:, 1] = s_ * a[:, 0] + c * a[:, 1] - 0.2
    b += torch.randn((npts, 2), dtype=torch.float64, device="cuda", generator=g) * 0.05
    icp = I.Icp3d(torch.zeros((1, 3), dtype=torch.float64, device="cuda"))
    icp.estimate_transform_device(a, b)  # warm-up: allocations, first-touch
    torch.cuda.synchronize()
    c0 = I.gn_path_counters(icp)
    t0 = time.perf_counter()
    _, applied = icp.estimate_transform_device(a, b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1 = I.gn_path_counters(icp)
    evals = sum(c1[k] - c0[k] for k in (0, 2, 3))  # evaluations launched (a window miss counts twice)
    icp.close()
    per_eval = dt / max(evals, 1)
    gbs = 96.0 * npts / per_eval / 1e9
    window = c1[0] > c0[0]
    digits = 2 if npts <= (4 << 20) else 3
    # bytes the launches really stream per point: residual pass 32 in + 16 out (the running sums ride with it since
    # round 3: no accumulate pass), every further selection pass 16 in.  Beyond 4M points the windows are refined in
    # two passes: the second histogram pass re-reads the residuals (16), the candidate pass then reads only the short
    # lists of residuals inside the fine windows (~1 % of the points).  The radix pipeline still accumulates last (32).
    refined = window and npts > (4 << 20)
    streamed = (48 + 16) if window else (48 + 16 * (2 * (digits + 1) - 1) + 32)
    return {"points": npts, "evaluations": int(evals), "inner_iterations": int(applied),
            "ms_per_evaluation": 1e3 * per_eval, "algorithmic_bytes_per_evaluation": 96 * npts,
            "achieved_GBs": gbs, "peak_GBs": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
            "streamed_bytes_per_point": streamed, "streamed_GBs": streamed * npts / per_eval / 1e9,
            "streamed_frac": streamed * npts / per_eval / 1e9 / HBM_PEAK_GBS,
            "pipeline": ("refined windows (statistics of a 256k-pair sample; residuals + histograms + block sums; second "
                         "histogram pass over the residuals; candidates from the fine-window lists, selection, fold)"
                         if refined else "window (2 launches)") if window
                        else f"radix digits ({2 * (digits + 1) + 1} launches)",
            "note": "reduce kernels alone on pairs past the 256 MiB Infinity Cache (SURVEY 8(d)(i)); bytes as SURVEY "
                    "prices them, selection passes not counted"}


def nn_large(npts, warm_searches=6):
    """the SEARCH kernel alone where HBM bandwidth is what it runs against -- N = M = `npts`
    (default 16M: target cloud + grid records + cell table + source snapshot are ~1.3 GB, five times the 256 MiB
    Infinity Cache; at the headline's 1M x 1M the same structures are 90 MB and sit in it).  Two estimate calls on
    device-generated clouds of the benchmark's shape -- one outer iteration (the seeded first search only) and
    1 + `warm_searches` -- timed by the library's HIP events around every search launch; their difference is the warm
    searches, the kernel of the headline's roofline line.  Bytes per launch as SURVEY prices them: 28 N + 24 M."""
    import torch
    import icp_rust_amd as I

    g = torch.Generator(device="cuda").manual_seed(4321)
    lo = torch.tensor([-40.0, -40.0, -2.0], dtype=torch.float64, device="cuda")
    hi = torch.tensor([40.0, 40.0, 6.0], dtype=torch.float64, device="cuda")
    # the box of the 1M pair scaled to keep its point density (2 targets per grid cell either way)
    scale = (npts / 1.0e6) ** (1.0 / 3.0)
    lo, hi = lo * scale, hi * scale

    def cloud(n):  # 70 % on the six faces of the box, 30 % inside it (synth.box_cloud, on the device)
        u = torch.rand((n, 4), dtype=torch.float64, device="cuda", generator=g)
        p = lo + u[:, 1:4] * (hi - lo)
        ext = hi - lo
        areas = torch.stack([ext[0] * ext[1]] * 2 + [ext[1] * ext[2]] * 2 + [ext[0] * ext[2]] * 2)  # +-z, +-x, +-y
        cum = torch.cumsum(areas, 0) / areas.sum()
        face = torch.searchsorted(cum, ((u[:, 0] - 0.3) / 0.7).clamp(0.0, 1.0).contiguous(), right=True).clamp(0, 5)
        on_face = u[:, 0] >= 0.3
        axis = torch.tensor([2, 2, 0, 0, 1, 1], device="cuda")[face]
        side = torch.tensor([0, 1, 0, 1, 0, 1], device="cuda")[face]
        val = torch.where(side == 1, hi[axis], lo[axis])
        rows = torch.nonzero(on_face).squeeze(1)
        p[rows, axis[rows]] = val[rows]
        return p.contiguous()

    dst = cloud(npts)
    src = cloud(npts)
    c, s_ = np.cos(0.015), np.sin(0.015)  # moved by the inverse of the truth pose of the 1M pair, + noise
    x, y = src[:, 0] - 0.3, src[:, 1] + 0.2
    src[:, 0] = c * x + s_ * y
    src[:, 1] = -s_ * x + c * y
    src += torch.randn((npts, 3), dtype=torch.float64, device="cuda", generator=g) * 0.01
    icp = I.Icp3d(dst)
    icp.estimate(src, I.Transform(), 2)  # warm-up: allocations, first touch
    icp.profile_enable(1)
    icp.profile_read()
    icp.estimate(src, I.Transform(), 1)
    ms1, k1 = icp.profile_read()
    _, _, inner = icp.estimate(src, I.Transform(), 1 + warm_searches, return_info=True)
    ms2, k2 = icp.profile_read()
    icp.profile_enable(0)
    engine = {I.NN_BRUTE: "brute", I.NN_GRID: "grid"}.get(I.lib().icp_get_nn_mode(icp._h), "?")
    icp.close()
    launches = int(k2 - k1)
    per = 1e-3 * (ms2 - ms1) / max(launches, 1)
    nn_bytes = 52.0 * npts
    gbs = nn_bytes / per / 1e9 if per > 0 else 0.0
    traffic, tf_src = None, None
    tf_name = next((f for f in ("r06_traffic_pmc_nn_large.json", "r05_traffic_pmc_nn_large.json")
                    if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
    if tf_name and npts == 16 * 1024 * 1024:
        traffic = json.load(open(os.path.join(ROOT, "profiles", tf_name)))["k_nn_grid"]["warm_kernel_bytes_per_launch"]  # (the line times warm searches)
        tf_src = f"profiles/{tf_name} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of profiles/nn_large_only.py)"
    return {"points": npts, "kernel": "k_nn_grid_warm_coop", "warm_searches": launches, "first_search_ms": ms1 / max(k1, 1),
            "ms_per_search": 1e3 * per, "algorithmic_bytes_per_launch": nn_bytes, "achieved_GBs": gbs, "peak_GBs": HBM_PEAK_GBS,
            "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tf_src,
            "traffic_ratio": None if not traffic else traffic / nn_bytes, "inner_iterations": [int(v) for v in inner],
            "engine": engine,
            "note": "the search kernel alone on clouds past the 256 MiB Infinity Cache (SURVEY 8(d)(ii)): target records, "
                    "cell table and source snapshot stream from HBM; timed by HIP events around each launch inside estimate()"}


def converging(calls, n, m, nn_mode):
    """Side line: the same 1M x 1M size on a pair that CONVERGES, in the regime of the reference's
    real scans (synth.converging_pair: millimetre coordinates, re-observed points + clutter), where the inner
    Gauss-Newton loops run from tens of updates per outer iteration down to none; the headline pair applies exactly
    one update per outer iteration.  Same call as the headline: estimate(src, I, 20)."""
    import torch
    import icp_rust_amd as I
    from icp_rust_amd import synth

    src, dst, truth_param = synth.converging_pair(n, m)
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    icp = I.Icp3d(d_dst, nn_mode=nn_mode)
    icp.estimate(d_src, I.Transform(), 3)
    torch.cuda.synchronize()
    skipped0 = I.fixed_point_skips(icp)
    t0 = time.perf_counter()
    for _ in range(calls):
        T, inner = icp.estimate(d_src, I.Transform(), MAX_ITER, return_info="inner")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # iterations the library did not run: once an outer iteration leaves the pose as it found it (bit for bit), the
    # ones after it repeat it, and only the last of them -- which reports the correspondences -- still runs; the
    # result (pose, indices, every inner count) is that of all twenty (tests/test_gpu_parity.py)
    skipped = I.fixed_point_skips(icp) - skipped0
    steps = calls * MAX_ITER
    steps_run = steps - skipped
    # ... and what the same call costs with ALL twenty iterations run: the handle's fixed-point exit
    # switched off (icp_set_fixed_point_exit, include/icp_mi355x_debug.h) -- same pose, indices and inner counts
    all_run = None
    if skipped > 0 and I.lib().icp_set_fixed_point_exit(icp._h, 0) == 0:
        Ta, inner_a = icp.estimate(d_src, I.Transform(), MAX_ITER, return_info="inner")
        torch.cuda.synchronize()
        s0 = I.fixed_point_skips(icp)
        t1 = time.perf_counter()
        for _ in range(calls):
            icp.estimate(d_src, I.Transform(), MAX_ITER)
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t1) / calls
        same = bool(np.array_equal(Ta.as_array(), T.as_array()) and np.array_equal(inner_a, inner))
        all_run = {"ms_per_call": 1e3 * t_all, "ms_per_step": 1e3 * t_all / MAX_ITER, "iterations_left_out": int(I.fixed_point_skips(icp) - s0),
                   "same_pose_and_inner_counts_as_the_default": same,
                   "how": "the handle's fixed-point exit switched off (icp_set_fixed_point_exit): every one of the twenty outer "
                          "iterations runs, also those that repeat the fixed point"}
        I.lib().icp_set_fixed_point_exit(icp._h, 1)
    icp.close()
    evals = [int(k) + 1 for k in inner]
    evals_run = (sum(evals) * calls - skipped) / max(steps_run, 1)  # (a skipped iteration would have been one evaluation)
    step_bytes = 28.0 * n + 24.0 * m + float(np.mean(evals)) * 96.0 * n
    truth = I.Transform(truth_param).as_array()
    return {"workload": f"converging pair, millimetres: src = {n} points, 70 % re-observed points of the {m}-point target cloud "
                        "moved by the inverse truth pose + noise, 30 % clutter (synth.converging_pair)",
            "gn_evaluations_per_step": float(np.mean(evals)),
            #  the rate counts the iterations that were RUN; per REQUESTED iteration -- what earlier rounds printed
            # as ms_per_step -- is kept under its own name, and `all_twenty_run` is the call with nothing left out
            "value": steps_run / dt, "unit": "iterations/s (iterations run)", "ms_per_step": 1e3 * dt / max(steps_run, 1),
            "steps": steps_run, "ms_per_call": 1e3 * dt / calls, "ms_per_requested_iteration": 1e3 * dt / steps,
            "all_twenty_run": all_run, "ms_per_step_all_run": all_run["ms_per_step"] if all_run else 1e3 * dt / steps,
            "fixed_point": {"iterations_requested": steps, "iterations_run": steps_run, "ms_per_iteration_run": 1e3 * dt / max(steps_run, 1),
                            "gn_evaluations_per_iteration_run": evals_run,
                            "note": "value and ms_per_step count the iterations that were RUN (ms_per_requested_iteration divides "
                                    "the call by the 20 it asks for, as earlier rounds' ms_per_step did); the "
                                    "iterations after the pose has stopped moving (inner count 0, pose bit-equal) repeat the "
                                    "one before them and are not run, except the last -- same pose, indices and inner counts "
                                    "as running all of them (icp_fixed_point_skips, include/icp_mi355x_debug.h)"},
            "inner_iterations_per_step": [int(k) for k in inner],
            "step_hbm": {"algorithmic_bytes_per_step": step_bytes, "achieved_GBs": step_bytes / (dt / steps) / 1e9,
                         "frac_of_8TBs": step_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBS},
            "pose_abs_err_vs_truth": float(np.max(np.abs(T.as_array() - truth)))}


def rotating(calls, n, m, nn_mode, d_dst):
    """Side line: consecutive estimate(src, T0, 20) calls on DIFFERENT clouds and starting poses --
    three source clouds (independent samples with different truth motions, one of them starting from a non-identity
    pose) alternate call by call against the headline's target cloud, so no call repeats the one before it: the
    per-call window predictions and whatever else a handle carries from call to call start from another cloud's
    statistics every time.  Same size, same call as the headline."""
    import torch
    import icp_rust_amd as I
    from icp_rust_amd import synth

    variants = [(synth.SEED + 11, (0.30, -0.20, 0.015), (0.0, 0.0, 0.0)), (synth.SEED + 23, (0.12, 0.25, -0.008), (0.0, 0.0, 0.0)),
                (synth.SEED + 37, (-0.22, 0.10, 0.011), (-0.05, 0.02, 0.002))]
    clouds = []
    for seed, param, init in variants:
        u = synth.uniforms(seed, n)
        sc = synth.box_cloud(seed, n, u=u)
        _, (c, sn, tx, ty) = synth._apply_se2(param, np.zeros((1, 2)))
        dx, dy = sc[:, 0] - tx, sc[:, 1] - ty
        sc[:, 0], sc[:, 1] = c * dx + sn * dy, -sn * dx + c * dy
        sc[:, 0] += synth.NOISE_SIGMA * np.sqrt(-2.0 * np.log(1.0 - u[:, 4])) * np.cos(2.0 * np.pi * u[:, 5])
        sc[:, 1] += synth.NOISE_SIGMA * np.sqrt(-2.0 * np.log(1.0 - u[:, 4])) * np.sin(2.0 * np.pi * u[:, 5])
        sc[:, 2] += synth.NOISE_SIGMA * np.sqrt(-2.0 * np.log(1.0 - u[:, 6])) * np.cos(2.0 * np.pi * u[:, 7])
        clouds.append((torch.from_numpy(np.ascontiguousarray(sc)).cuda(), I.Transform(list(init))))
    icp = I.Icp3d(d_dst, nn_mode=nn_mode)
    for d_s, T0 in clouds:  # (allocations, first statistics)
        icp.estimate(d_s, T0, 2)
    per_call, inner_all = [], []
    for k in range(calls):
        d_s, T0 = clouds[k % len(clouds)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T, inner = icp.estimate(d_s, T0, MAX_ITER, return_info="inner")
        torch.cuda.synchronize()
        per_call.append(time.perf_counter() - t0)
        inner_all.append([int(x) for x in inner])
    icp.close()
    med = float(np.median(per_call))
    return {"workload": f"{len(clouds)} different {n}-point source clouds (own seeds, own truth motions, one starting from a "
                        f"non-identity pose) alternating call by call against the headline's {m}-point target cloud; "
                        "estimate(src_k, T0_k, 20) each",
            "calls": calls, "ms_per_step": 1e3 * med / MAX_ITER, "value": MAX_ITER / med, "unit": "iterations/s",
            "ms_per_step_per_call": [1e3 * t / MAX_ITER for t in per_call],
            "inner_iterations_per_step_per_cloud": inner_all[:len(clouds)]}


def reference_sized():
    """Part of the CPU-baseline leg.  BASELINE configs[0] / configs[1] at the reference's own sizes,
    next to the single-thread CPU oracle: a 650-point 2-D scan pair of the reference's scans/2d and a 28.8k-point 3-D frame in the
    scans.hdf5 packet layout (synthetic stand-in: the file is absent from the reference mount), each as
    one estimate(src, identity, 20) from host buffers -- what examples/scan2d / scan3d call per frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import icp_rust_amd as I
    import oracle_ffi as O
    from icp_rust_amd import synth
    from icp_rust_amd.scans import load_scan2d

    def timed(f, reps):
        f()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        return 1e3 * (time.perf_counter() - t0) / reps

    out = {}
    g = os.path.join(ROOT, "tests", "golden", "scans2d")
    s2, d2 = load_scan2d(os.path.join(g, "001.txt")), load_scan2d(os.path.join(g, "002.txt"))
    icp2 = I.Icp2d(d2)
    tree2 = O.KdTree(d2)
    out["scan2d_pair"] = {"points": [len(s2), len(d2)],
                          "gpu_ms_per_estimate20": timed(lambda: icp2.estimate(s2, I.Transform(), 20), 10),
                          "cpu_oracle_ms_per_estimate20": timed(lambda: tree2.estimate(s2, O.transform_identity(), 20), 5),
                          "note": "inner counts 11, 8, 9, 6, 3, 1, 1, 1, 0, 0, ...: the iterations after the pose has stopped "
                                  "moving repeat the ninth and are not run, except the last (same result as all twenty)"}
    icp2.close()
    pk = synth.synthetic_scan3d_packets(150)
    s3, d3 = synth.remove_invalid_values(pk[:75]), synth.remove_invalid_values(pk[75:150])
    icp3 = I.Icp3d(d3)
    tree3 = O.KdTree(d3)
    out["scan3d_frame"] = {"points": [len(s3), len(d3)],
                           "gpu_ms_per_estimate20": timed(lambda: icp3.estimate(s3, I.Transform(), 20), 10),
                           "cpu_oracle_ms_per_estimate20": timed(lambda: tree3.estimate(s3, O.transform_identity(), 20), 1)}
    icp3.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-src", type=int, default=1_000_000)
    ap.add_argument("--n-dst", type=int, default=1_000_000)
    ap.add_argument("--nn", choices=["auto", "brute", "grid"], default="auto",
                    help="NN engine of the headline value (auto = the library default: exact grid search "
                         "at this size); all engines return bit-identical correspondences")
    ap.add_argument("--brute-steps", type=int, default=3,
                    help="outer iterations of the brute-force sweep measured alongside (0 = skip)")
    ap.add_argument("--weak-steps", type=int, default=40,
                    help="N > 1: outer iterations of the weak-scaling line (N x n-src source points; 0 = skip)")
    ap.add_argument("--cpu-iters", type=int, default=10,
                    help="outer iterations of the CPU baseline: ~7 s on one core + ~3 s with all cores (0 = skip)")
    ap.add_argument("--converging-calls", type=int, default=3,
                    help="estimate(20) calls on the CONVERGING 1M pair (src re-observes points of dst: inner loops of "
                         "several updates, the regime of the reference's real scans) timed as a side line (0 = skip)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region of K steps is repeated this many times (each bracketed by barrier + synchronize); "
                         "value = K / the MEDIAN region (a 3 ms region alone is +-3 % run to run)")
    ap.add_argument("--rotating-calls", type=int, default=9,
                    help="estimate(20) calls of the rotating-inputs side line: three different clouds / poses alternating (0 = skip)")
    ap.add_argument("--nn-points", type=int, default=16 * 1024 * 1024,
                    help="points of each cloud of the `nn_large` line: the search kernel past the Infinity Cache (0: skip)")
    ap.add_argument("--gn-points", type=int, default=64 * 1024 * 1024,
                    help="pairs of the separate 'reduce kernels alone, past the Infinity Cache' line (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import icp_rust_amd as I
    from icp_rust_amd import synth
    from icp_rust_amd.dist import BlockShardedIcp, HipStages, ShardedIcp, TorchComm, block_shard, shard_range

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # ICP_BENCH_SHARE_GPU=1 (debugging on a 1-GPU box only): all ranks use cuda:0 and talk gloo,
    # to exercise the sharded path end to end; never set for a measurement
    share_gpu = os.environ.get("ICP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if share_gpu:
            import datetime

            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=240))
        else:
            # (a collective that some rank never enters must end the run with an error within minutes, not hang it)
            import datetime

            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=240))

    I.build()
    n, m = args.n_src, args.n_dst
    lo, hi = shard_range(n, rank, world)
    src_np, dst_np = synth.synthetic_pair(n, m, src_first=lo, src_count=hi - lo)
    d_dst = torch.from_numpy(dst_np).cuda()
    d_src = torch.from_numpy(src_np).cuda()  # contiguous shard: what the sweep engine's driver takes
    d_src_full = None
    if world > 1:
        # every rank regenerates the whole source cloud (counter-based generator: no communication) and
        # keeps the points of its reduction-tree blocks (dist.BlockShardedIcp.take_source)
        full_np, _ = synth.synthetic_pair(n, 1)
        d_src_full = torch.from_numpy(full_np).cuda()
    nn_mode = {"auto": I.NN_AUTO, "brute": I.NN_BRUTE, "grid": I.NN_GRID}[args.nn]
    comm = TorchComm(rank, world) if world > 1 else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loop_state = {"mode": None}

    def measure(mode, steps, warmup, want_parity=False, weak=False, repeats=1):
        """K timed outer iterations = one Icp3d::estimate(src, T, K) call on resident data.
        N ranks: the grid engine runs block-sharded (dist.BlockShardedIcp: every rank searches and
        evaluates the points of its reduction-tree blocks, two small exchanges per evaluation, same
        bits as one GPU); the sweep, whose search is 99.8 % of the step, keeps contiguous shards and a
        replicated inner loop (dist.ShardedIcp).  weak: N x n source points instead of n."""
        icp = I.Icp3d(d_dst, device=local_rank, nn_mode=mode)
        T = I.Transform()
        n_run = n * world if weak else n
        block = world > 1 and mode != I.NN_BRUTE
        if world == 1:
            # one rank: the handle keeps its own streams (the library overlaps the next search with the
            # evaluation that decides it); nothing else is enqueued on them
            if warmup > 0:
                icp.estimate(d_src, T, warmup)
        elif block:
            full = d_src_full
            if weak:
                full = torch.from_numpy(synth.synthetic_pair(n_run, 1)[0]).cuda()
            driver = BlockShardedIcp({rank: HipStages(icp)}, n_run, world, comm)
            # The inner loops as one launch per rank, exchanging through hipIpc-mapped inboxes (gn_loop.hip); the stage calls
            # + collectives serve whatever a launch hands back.  ICP_DIST_NO_LOOP=1: stage calls only.  The path has run
            # between processes on ONE GPU only (tests/test_gpu_ipc.py): the warm-up doubles as its probation -- if any rank
            # fails to connect or a launch gives up waiting for a peer, EVERY rank falls back to the stage calls.
            # every call: fold order of the whole cloud (the sort one GPU does per call), this rank's blocks out
            # of it, then the iterations -- all inside the timed region, like the one-GPU call's snapshot
            loop_state["mode"] = "stage calls + collectives (ICP_DIST_NO_LOOP=1)"
            if os.environ.get("ICP_DIST_NO_LOOP") != "1":
                ok, why, transport = 1, "", None
                try:
                    # the first transport whose ping-pong probe passes on every rank: device memory through hipIpc for
                    # ranks that share a device, fine-grained device memory through hipIpc, pinned host memory in a
                    # shared-memory object (dist.BlockShardedIcp.connect_loop); none: the stage calls serve
                    transport = driver.connect_loop()  # (collective; agrees on its outcome itself)
                    if transport is None:
                        ok, why = 0, "no transport passed the ping-pong probe"
                except Exception as e:  # noqa: BLE001
                    ok, why = 0, repr(e)
                #  every step below that issues collectives is entered by ALL ranks or by none: the ranks agree on
                # `ok` before each of them, so a rank with a local failure cannot leave its peers inside a collective
                ok = 1 if comm.all_ok(ok == 1) else 0
                if ok:
                    try:
                        # (the driver agrees on the outcome of every launch that exchanges through the inboxes and restarts
                        # the call through the stage calls if a wait ran out anywhere: dist.BlockShardedIcp.estimate)
                        T, k_loop, _ = driver.estimate_full(full, T, max(warmup, 2))  # (also seeds the window predictions)
                        if driver.counters.get("loop_gave_up", 0):
                            ok, why = 0, "a launch gave up waiting for a peer"
                    except Exception as e:  # noqa: BLE001
                        ok, why = 0, repr(e)
                    ok = 1 if comm.all_ok(ok == 1) else 0
                if ok:
                    try:
                        #  ... and the SAME call through the stage calls + collectives: pose and inner counts
                        # must be the launches', bit for bit -- a transport that delivered stale words which still
                        # passed the kernels' count cross-checks would show here, before anything is timed
                        saved, driver._loop = driver._loop, None
                        T_chk, k_chk, _ = driver.estimate_full(full, I.Transform(), max(warmup, 2))
                        driver._loop = saved
                        if T_chk.as_array().tobytes() != T.as_array().tobytes() or not np.array_equal(k_chk, k_loop):
                            ok, why = 0, "the launches' result differs from the stage calls' on the same call"
                    except Exception as e:  # noqa: BLE001
                        ok, why = 0, repr(e)
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 1:
                    loop_state["mode"] = f"one launch per rank and inner loop, inboxes: {transport}"
                    loop_state["transport"] = transport
                else:
                    loop_state["mode"] = "stage calls + collectives (the one-launch path failed its probation: " + (why or "on another rank") + ")"
                    try:
                        driver.disconnect_loop()
                    except Exception:  # noqa: BLE001
                        pass
                    icp.close()
                    icp = I.Icp3d(d_dst, device=local_rank, nn_mode=mode)
                    driver = BlockShardedIcp({rank: HipStages(icp)}, n_run, world, comm)
                    T = I.Transform()
            if "one launch" not in loop_state["mode"]:
                T, _, _ = driver.estimate_full(full, T, max(warmup, 2))  # (also seeds the window predictions)
        else:
            driver = ShardedIcp(HipStages(icp), n, rank, world, src_full=d_src_full)
            driver.stages.prepare(d_src, T)
            for _ in range(warmup):
                T, _ = driver.step(d_src, T)
        # HIP events around every search launch of a short run (<= 40 steps: the cold first search of
        # each 20-step estimate call then weighs exactly its 1-in-20 share, as in a rocprofv3 trace of
        # the same command); on longer runs around every 7th launch (an event pair costs a few us of
        # stream time; 7 is coprime with the 20-step cycle, so cold searches are sampled at their share)
        icp.profile_enable(1 if steps <= 40 else 7)
        icp.profile_read()
        regions, inner = [], []
        for _rep in range(max(repeats, 1)):
            barrier()
            t0 = time.perf_counter()
            inner = []
            if world == 1:
                # one rank: the library's own outer loop (icp_estimate_device), one call per 20 steps as
                # examples/scan3d.rs:131 issues per frame; its per-call setup (cell-sorted snapshot of
                # the source cloud) is timed too
                done = 0
                while done < steps:
                    k_iters = min(MAX_ITER, steps - done)
                    T, k = icp.estimate(d_src, I.Transform(), k_iters, return_info="inner")
                    inner.extend(int(x) for x in k[:k_iters])
                    done += k_iters
            elif block:
                done = 0
                while done < steps:
                    k_iters = min(MAX_ITER, steps - done)
                    T, k, _ = driver.estimate_full(full, I.Transform(), k_iters)
                    inner.extend(int(x) for x in k[:k_iters])
                    done += k_iters
            else:
                # N ranks, sweep engine: stage calls around one index all-gather per iteration
                for k_step in range(steps):
                    if k_step % MAX_ITER == 0:
                        T = I.Transform()
                        driver.stages.prepare(d_src, T)
                    T, k = driver.step(d_src, T)
                    inner.append(int(k))
            barrier()
            regions.append(time.perf_counter() - t0)
        if world > 1:  # every region: the slowest rank's
            t = torch.tensor(regions, dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            regions = [float(x) for x in t.tolist()]
        elapsed = float(np.median(regions))
        nn_ms, nn_launches = icp.profile_read()
        icp.profile_enable(0)
        alone_ms = None
        if world == 1 and steps >= 20:
            # outside the timed region: the same search kernel with nothing beside it (stage calls on one
            # stream: no speculative overlap), for comparison with the overlapped duration above
            solo = ShardedIcp(HipStages(icp), n, 0, 1)
            Ts = I.Transform()
            solo.stages.prepare(d_src, Ts)
            for _ in range(3):
                Ts, _ = solo.step(d_src, Ts)
            icp.profile_enable(1)
            icp.profile_read()
            for _ in range(12):
                Ts, _ = solo.step(d_src, Ts)
            a_ms, a_n = icp.profile_read()
            icp.profile_enable(0)
            alone_ms = a_ms / max(a_n, 1)
        engine = {I.NN_BRUTE: "brute", I.NN_GRID: "grid"}[I.lib().icp_get_nn_mode(icp._h)]
        checked = None
        if world == 1 and want_parity:
            # outside the timed region: the same call once more with the index buffer handed back, for the
            # comparison with the oracle in the cpu_baseline leg
            k_last = min(MAX_ITER, steps)
            checked = (icp.estimate(d_src, I.Transform(), k_last, return_info=True), k_last, icp.last_fold_order(n))
        counters = dict(driver.counters) if block else None
        if block and getattr(driver, "_loop", None) is not None:
            driver.disconnect_loop()  # (the peers' mappings go back before any rank pools its handle)
        icp.close()
        return dict(elapsed=elapsed, regions=regions, steps=steps, inner=inner, nn_ms=nn_ms, nn_launches=nn_launches, T=T,
                    engine=engine, alone_ms=alone_ms, checked=checked, counters=counters, n_run=n_run)

    def nn_roofline(r, n_shard):
        """Roofline of the dominant kernel (the NN search) from the live HIP-event timing."""
        avg_s = 1e-3 * r["nn_ms"] / max(r["nn_launches"], 1)
        # compulsory bytes 24 B/source in + 4 B idx out + 24 B/target in (once)
        nn_bytes = 28.0 * n_shard + 24.0 * m
        gbs = nn_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        if r["engine"] == "brute":
            # 3 sub, 3 mul, 2 add per (source, target) pair.  The sweep screens every pair in f32 and
            # re-evaluates only possible winners/ties in f64 (~ln M per query), so the arithmetic that
            # bounds it is the FP32 vector rate.
            flops = 8.0 * n_shard * m
            tf = flops / avg_s / 1e12 if avg_s > 0 else 0.0
            return {"kernel": "k_nn_brute_dot", "bound": "fp32_valu", "achieved": tf, "peak": FP32_VALU_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": tf / FP32_VALU_PEAK_TFLOPS, "traffic": None,
                    "avg_launch_ms": 1e3 * avg_s, "launches": int(r["nn_launches"]),
                    "algorithmic_flops_per_launch": flops,
                    "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "algorithmic_bytes_per_launch": nn_bytes}}
        # `traffic`: fabric-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, collected in
        # separate passes and committed under profiles/ (a PMC pass cannot run inside this process); only
        # valid for the full-size single-GPU workload it was measured on
        traffic, tf_name = None, None
        for tf_name in ("r06_traffic_pmc.json", "r05_traffic_pmc.json", "r04_traffic_pmc.json", "r03_traffic_pmc.json", "r02_traffic_pmc.json", "r01_traffic_pmc.json"):
            tf_path = os.path.join(ROOT, "profiles", tf_name)
            if os.path.exists(tf_path):
                break
        if os.path.exists(tf_path) and world == 1 and n == 1_000_000 and m == 1_000_000:
            traffic = json.load(open(tf_path))["k_nn_grid"]["traffic_bytes_per_launch"]
        return {"kernel": "k_nn_grid_warm_coop (one search = one launch; the first search of a call = k_nn_grid_seeded: seeds + the same walk)", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": f"profiles/{tf_name} (separate rocprofv3 --pmc passes of this command; a PMC pass "
                                  "cannot run inside the timed process)",
                "avg_launch_ms": 1e3 * avg_s,
                "launches": int(r["nn_launches"]), "algorithmic_bytes_per_launch": nn_bytes,
                # in the timed region the kernel shares the CUs with the Gauss-Newton evaluation that decides
                # its pose (speculative search); alone it is shorter:
                "alone": None if not r.get("alone_ms") else {
                    "avg_launch_ms": r["alone_ms"], "achieved": nn_bytes / (1e-3 * r["alone_ms"]) / 1e9,
                    "frac": nn_bytes / (1e-3 * r["alone_ms"]) / 1e9 / HBM_PEAK_GBS}}

    res = measure(nn_mode, args.steps, args.warmup, want_parity=args.cpu_iters > 0, repeats=args.repeats)
    brute = None
    if args.brute_steps > 0 and res["engine"] != "brute":
        brute = measure(I.NN_BRUTE, args.brute_steps, 1, want_parity=args.cpu_iters > 0)
    weak = None
    if world > 1 and res["engine"] == "grid" and args.weak_steps > 0:
        weak = measure(nn_mode, args.weak_steps, 2, weak=True)
    elapsed, inner, T = res["elapsed"], res["inner"], res["T"]

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        n_shard = block_shard(n, rank, world)[3] if (world > 1 and res["engine"] == "grid") else hi - lo
        truth = I.Transform(synth.TRUTH_PARAM).as_array()
        evals = [k + 1 for k in inner]  # GN evaluations per step: the applied updates + the terminating one
        # whole-step algorithmic HBM bytes: 28 N + 24 M + k * 96 N
        step_bytes = 28.0 * n + 24.0 * m + float(np.mean(evals)) * 96.0 * n
        out = {
            "metric": "ICP iterations/sec on 1M-pt 3D pair",
            "value": args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "timed_regions": {"repeats": len(res["regions"]), "steps_each": args.steps, "value_from": "median region",
                              "ms_per_step_each": [1e3 * r / args.steps for r in res["regions"]]},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic 3D 1M-vs-1M point clouds, exact NN + Huber/MAD Gauss-Newton, "
                            "point-to-point SE(2)-on-xy ICP (BASELINE.json configs[2]; sharded = configs[3]); "
                            "NN engine = " + res["engine"] + " (bit-identical correspondences to the "
                            "brute-force sweep, which is timed alongside under `brute_force`)",
                "n_src": n, "n_dst": m, "nn": res["engine"], "outer_iterations_per_estimate_call": MAX_ITER,
                "parallelism": ("one GPU" if world == 1 else
                                f"source cloud sharded x{world} by reduction-tree block (every rank searches and evaluates "
                                "its blocks' points with the one-GPU pipeline -- search, paired first launches, finishing "
                                "workgroups -- and the ranks meet inside the finishing workgroups: window histograms, block sums "
                                "and candidates through mapped inboxes over xGMI, two exchanges per evaluation, the next search "
                                "already enqueued behind them; RCCL serves the rendezvous, one agreement flag per call and "
                                "whatever the pipeline hands back), target replicated; bit-identical to one GPU"
                                if res["engine"] == "grid" else
                                f"NN over contiguous source shards x{world} (index all-gather), target replicated, inner "
                                "loop replicated"),
                "seed": hex(synth.SEED),
            },
            "roofline": nn_roofline(res, n_shard),
            "step_hbm": {"algorithmic_bytes_per_step": step_bytes,
                         "achieved_GBs": step_bytes / (elapsed / args.steps) / 1e9,
                         "frac_of_8TBs": step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
            "inner_iterations_per_step": inner,
            "nn_share_of_step": (res["nn_ms"] / max(res["nn_launches"], 1)) / ms_per_step,
            "pose": T.as_array().tolist(),
            "pose_abs_err_vs_truth": float(np.max(np.abs(T.as_array() - truth))),
        }
        if res.get("counters"):
            out["sharded_evaluations"] = dict(res["counters"], inner_loops=loop_state["mode"], transport=loop_state.get("transport"),
                                              rccl_world=dist.get_world_size() if dist.is_initialized() else 1,
                                              backend=dist.get_backend() if dist.is_initialized() else None)
            out["n_gpus_note"] = ("strong scaling of ONE 1M-point registration (BASELINE configs[3]): the step is a chain of dependent "
                                  "launches of ~25-50 us each, so more GPUs shorten only the search and the first launches; "
                                  "`weak_scaling` (N x 1M source points) is where N GPUs pay, `brute_force` is configs[3] as worded")
        rccl_world = dist.get_world_size() if dist.is_initialized() else 1
        backend = dist.get_backend() if dist.is_initialized() else None
        #  every multi-GPU line says how its ranks talked: the headline line too
        out["transport"] = (loop_state.get("transport") or loop_state["mode"]) if (world > 1 and res.get("counters")) else None
        out["rccl_world"], out["backend"] = rccl_world, backend
        if brute is not None:
            out["brute_force"] = {
                "workload": "BASELINE configs[2] / [3] as worded: brute-force NN (LDS-tiled sweep) + Huber, 1M x 1M; N ranks: the "
                            "search over contiguous source shards (99.8 % of the step), one index all-gather per iteration, "
                            "the inner loop replicated",
                "value": brute["steps"] / brute["elapsed"], "unit": "iterations/s", "steps": brute["steps"],
                "ms_per_step": 1e3 * brute["elapsed"] / brute["steps"], "roofline": nn_roofline(brute, hi - lo),
                "scaling": "strong", "n_gpus": world, "transport": (f"{'RCCL' if backend == 'nccl' else backend} all_gather (torch.distributed)") if world > 1 else None,
                "rccl_world": rccl_world, "backend": backend,
            }
        if weak is not None:
            wv = weak["steps"] / weak["elapsed"]
            out["weak_scaling"] = {
                "workload": f"{weak['n_run']} source points ({world} x {n}) against the {m}-point target, grid engine, "
                            "block-sharded", "n_gpus": world, "scaling": "weak",
                "value": wv, "unit": "iterations/s", "steps": weak["steps"], "ms_per_step": 1e3 / wv,
                "source_points_per_second": wv * weak["n_run"],
                "inner_iterations_per_step": weak["inner"], "sharded_evaluations": weak["counters"],
                "transport": loop_state.get("transport"), "inner_loops": loop_state["mode"], "rccl_world": rccl_world, "backend": backend,
                "tree_blocks": block_shard(weak["n_run"], rank, world)[2],
                "note": "compare source_points_per_second with n_src x value of the 1-GPU line: every rank searches and evaluates "
                        f"its own {n} points with the launches one GPU runs on {n} (the reduction tree grows with the cloud: a rank "
                        "owns 256 of its blocks), and the ranks meet in the finishing workgroups of the evaluations",
            }
        if world == 1 and args.rotating_calls > 0:
            out["rotating_inputs"] = rotating(args.rotating_calls, n, m, nn_mode, d_dst)
        if world == 1 and args.converging_calls > 0:
            out["converging_pair"] = converging(args.converging_calls, n, m, nn_mode)
        if world == 1 and args.gn_points > 0:
            out["gn_large"] = gn_large(args.gn_points)
        if world == 1 and args.nn_points > 0 and nn_mode != I.NN_BRUTE:
            out["nn_large"] = nn_large(args.nn_points)
        if world == 1 and args.cpu_iters > 0:
            par = None
            if res["checked"] is not None:
                blocks, threads = I.reduce_geometry(n)
                par = (res["checked"][0], res["checked"][1], blocks, threads, res["checked"][2])
            out["cpu_baseline"] = cpu_baseline(src_np, dst_np, args.cpu_iters, parity_of=par)
            out["parity"] = out["cpu_baseline"].pop("parity")
            if brute is not None and brute.get("checked") is not None:
                # the sweep at its own size: the same estimate(steps) against the oracle in device order
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_ffi as O
                bg, bk, bperm = brute["checked"]
                bb, bt = I.reduce_geometry(n)
                out["brute_force"]["parity"] = parity_vs_oracle(O.KdTree(dst_np), src_np, bg, bk, bb, bt, bperm)
            # (the CPU side of these is the oracle too: part of the same baseline leg)
            out["cpu_baseline"]["reference_sized"] = reference_sized()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
