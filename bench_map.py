#!/usr/bin/env python3
"""EXTENSION beyond the reference (BASELINE.json configs[4], scan-to-growing-map; include/icp_mi355x.h
section 6): frames of 28 800 points registered with the reference's estimator (Icp3d::estimate, 20
iterations, warm-started) against a map of more than 10 M points that every registered frame is
appended to.  One GPU; the map is device resident.  Not the headline benchmark (bench.py) -- there is
no reference number for it.  `--point-to-plane K` adds the same loop with the point-to-plane residual
(the other labelled extension: normals from K nearest map points, computed when a point is inserted).

Prints one JSON line: per-frame registration time, per-frame append time (= rebuilding the search
grid over the whole map), map build time, frames/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--frame-points", type=int, default=75 * 384)
    ap.add_argument("--max-iter", type=int, default=20)
    ap.add_argument("--virtual-ranks", type=int, default=0, metavar="W",
                    help="also run the point-to-point loop through icp_create_multi with W ranks on cuda:0 (the N-rank "
                         "path of configs[4] rehearsed on one GPU; 0 = skip)")
    ap.add_argument("--point-to-plane", type=int, default=0, metavar="K",
                    help="also run the loop with point-to-plane residuals, normals from K nearest map points (0 = skip)")
    args = ap.parse_args()

    import torch

    import icp_rust_amd as I
    from icp_rust_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench_map.py needs a GPU: the product path has no CPU fallback")
    I.build()
    m0 = args.map_points
    chunk = 1 << 20
    d_map = torch.empty((m0, 3), dtype=torch.float64, device="cuda")
    for first in range(0, m0, chunk):
        cnt = min(chunk, m0 - first)
        d_map[first:first + cnt] = torch.from_numpy(synth.box_cloud(synth.SEED + 200, cnt, first=first)).cuda()
    motion = np.array([0.02, -0.01, 0.001])
    scans = []
    for k in range(1, args.frames + 1):
        # a fresh sample of the same world, seen from the sensor pose Exp(k * motion), with sensor noise
        s, _ = synth.synthetic_pair(args.frame_points, 1, seed=synth.SEED + 300 + 2 * k, param=tuple(k * motion))
        scans.append(torch.from_numpy(s).cuda())
    torch.cuda.synchronize()

    t0 = time.perf_counter()
    world = I.Icp3d(d_map)
    world.synchronize()
    t_build = time.perf_counter() - t0
    world.reserve(m0 + (args.frames + 1) * args.frame_points)
    del d_map
    # untimed: first-call allocations (workspace for one frame)
    world.estimate(scans[0], I.Transform(), 1)

    T = I.Transform()
    est_ms, app_ms, inner_all, err = [], [], [], []
    for k, scan in enumerate(scans, start=1):
        t0 = time.perf_counter()
        T, inner = world.estimate(scan, T, args.max_iter, return_info="inner")
        t1 = time.perf_counter()
        world.append(scan, T)
        t2 = time.perf_counter()
        est_ms.append(1e3 * (t1 - t0))
        app_ms.append(1e3 * (t2 - t1))
        inner_all.append(int(inner.sum()))
        truth = I.Transform(tuple(k * motion)).as_array()
        err.append(float(np.max(np.abs(T.as_array() - truth))))
    frame_ms = float(np.mean(est_ms) + np.mean(app_ms))
    out = {
        "metric": "scan-to-growing-map frames/s (EXTENSION: not a reference workload)",
        "value": 1e3 / frame_ms, "unit": "frames/s", "n_gpus": 1,
        "config": {"workload": "BASELINE.json configs[4] stand-in: synthetic frames registered against a growing map, "
                               "point-to-point (the reference's estimator; the point-to-plane extension: tests/test_p2plane.py)",
                   "map_points_start": m0, "map_points_end": world.target_count,
                   "frame_points": args.frame_points, "frames": args.frames, "max_iter": args.max_iter},
        "ms_per_frame": frame_ms,
        "estimate_ms": {"mean": float(np.mean(est_ms)), "min": float(np.min(est_ms)), "max": float(np.max(est_ms))},
        "append_ms": {"mean": float(np.mean(app_ms)), "min": float(np.min(app_ms)), "max": float(np.max(app_ms)),
                      "note": "transform + append + search-grid update: the sorted records move up by their cells' shifts and the new ones "
                              "fill the gaps (icp_grid_append_counters: incremental / rebuilt appends below); a full rebuild "
                              "only for points beyond half a cell outside the grid's box or once the map has grown by half"},
        "appends_incremental_rebuilt": list(world.append_counters()),
        "map_build_ms": 1e3 * t_build,
        "inner_updates_per_frame": inner_all,
        "pose_abs_err_vs_truth_last_frame": err[-1],
        "dtype": "f64", "data": "synthetic",
    }
    world.close()
    if args.virtual_ranks:
        # the same loop across W ranks (replicated map, the scan's points sharded by reduction-tree block): host
        # buffers, every rank appends the registered scan to its replica (icp_multi_append_targets)
        W = args.virtual_ranks
        host_map = np.concatenate([synth.box_cloud(synth.SEED + 200, min(chunk, m0 - f), first=f) for f in range(0, m0, chunk)])
        t0 = time.perf_counter()
        multi = I.IcpMulti(host_map, [0] * W)
        t_build_multi = time.perf_counter() - t0
        del host_map
        host_scans = [s.cpu().numpy() for s in scans]
        multi.estimate(host_scans[0], I.Transform(), 1)
        T = I.Transform()
        est, app, merr = [], [], []
        for k, scan in enumerate(host_scans, start=1):
            t0 = time.perf_counter()
            T = multi.estimate(scan, T, args.max_iter)
            t1 = time.perf_counter()
            multi.append(scan, T)
            t2 = time.perf_counter()
            est.append(1e3 * (t1 - t0))
            app.append(1e3 * (t2 - t1))
            merr.append(float(np.max(np.abs(T.as_array() - I.Transform(tuple(k * motion)).as_array()))))
        out["virtual_ranks"] = {
            "ranks": W, "map_points_end": multi.target_count, "estimate_ms": float(np.mean(est)),
            "append_ms": float(np.mean(app)), "ms_per_frame": float(np.mean(est) + np.mean(app)),
            "create_ms": 1e3 * t_build_multi, "pose_abs_err_vs_truth_last_frame": merr[-1],
            "sharded_replicated_evaluations": list(multi.counters()),
            "note": "icp_create_multi with every rank on cuda:0, host buffers: a functional rehearsal of the N-rank loop "
                    "(W replicas of the map on one GPU, W grid rebuilds per append), not a measurement of N GPUs"}
        multi.close()
    if args.point_to_plane:
        # the same frames against a fresh copy of the map, registered point-to-plane
        k_nn = args.point_to_plane
        d_map = torch.empty((m0, 3), dtype=torch.float64, device="cuda")
        for first in range(0, m0, chunk):
            cnt = min(chunk, m0 - first)
            d_map[first:first + cnt] = torch.from_numpy(synth.box_cloud(synth.SEED + 200, cnt, first=first)).cuda()
        world = I.Icp3d(d_map)
        world.reserve(m0 + (args.frames + 1) * args.frame_points)
        del d_map
        world.synchronize()
        t0 = time.perf_counter()
        world.compute_normals(k_nn)
        t_normals = time.perf_counter() - t0
        world.estimate_point_to_plane(scans[0], I.Transform(), 1)  # untimed: first-call allocations
        T = I.Transform()
        est, app, upd, perr = [], [], [], []
        for k, scan in enumerate(scans, start=1):
            t0 = time.perf_counter()
            T = world.estimate_point_to_plane(scan, T, args.max_iter)
            t1 = time.perf_counter()
            world.append(scan, T)
            t2 = time.perf_counter()
            world.update_normals(k_nn)
            t3 = time.perf_counter()
            est.append(1e3 * (t1 - t0))
            app.append(1e3 * (t2 - t1))
            upd.append(1e3 * (t3 - t2))
            perr.append(float(np.max(np.abs(T.as_array() - I.Transform(tuple(k * motion)).as_array()))))
        out["point_to_plane"] = {
            "k": k_nn, "normals_of_the_initial_map_ms": 1e3 * t_normals,
            "estimate_ms": float(np.mean(est)), "append_ms": float(np.mean(app)),
            "normals_of_the_appended_points_ms": float(np.mean(upd)),
            "ms_per_frame": float(np.mean(est) + np.mean(app) + np.mean(upd)),
            "pose_abs_err_vs_truth_last_frame": perr[-1],
            "note": "labelled extension, no reference behaviour: normals at insertion time (icp_update_target_normals)"}
        world.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
