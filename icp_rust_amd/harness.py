"""Headless frame loops of the reference's two examples: same file
handling, same warm start, same per-frame `Icp::new` + `estimate(src, transform, 20)`, the
trajectory instead of a window.

`icp_factory(dst) -> object with .estimate(src, Transform, max_iter) -> Transform` defaults to
the GPU classes; tests inject an oracle-backed factory to check the loop semantics on CPU.
"""
import os

import numpy as np

from .api import Icp2d, Icp3d, Transform
from .scans import load_scan2d
from .synth import PACKETS_PER_FRAME, remove_invalid_values


def run_scan2d(scan_dir, max_iter=20, icp_factory=None, max_frames=None):
    """examples/scan2d.rs:62-115.  `index` starts at 0 and is incremented BEFORE use, so
    000.txt is never read and 001.txt is the fixed source (:63,69-77); every later frame k
    loads dst = k.txt, builds Icp2d::new(&dst) and estimates warm-started from the previous
    frame (:85-88); the plotted pose is transform.inverse() (:90) and its translation is
    appended to the path (:105).  The loop ends at the first missing file (:72).
    Returns (transforms, inverse_transforms, path_xy)."""
    icp_factory = icp_factory or Icp2d
    index = 0
    src = None
    transform = Transform.identity()
    transforms, inverses, path = [], [], []
    while max_frames is None or len(transforms) < max_frames:
        index += 1
        filename = os.path.join(scan_dir, f"{index:03d}.txt")
        if not os.path.exists(filename):
            break
        if index == 1:
            src = load_scan2d(filename)
            continue
        dst = load_scan2d(filename)
        icp = icp_factory(dst)
        transform = icp.estimate(src, transform, max_iter)
        inv = transform.inverse()
        transforms.append(transform)
        inverses.append(inv)
        path.append(inv.t.copy())
    return transforms, inverses, np.array(path).reshape(-1, 2)


def run_scan3d(packets, step=PACKETS_PER_FRAME, max_iter=20, icp_factory=None):
    """examples/scan3d.rs:104-158 on an in-memory packet array (n_packets, 384, 3): the source
    is packets [0, step) with invalid returns removed (:104-107, 63-69); every frame takes
    dst = packets [index, index+step), filtered, THEN advances index (:113-121) -- so the first
    frame registers the source against itself; Icp3d::new(&dst); estimate warm-started
    (:130-131); path of transform.inverse().t (:133,144).
    Returns (transforms, inverse_transforms, path_xy)."""
    icp_factory = icp_factory or Icp3d
    packets = np.asarray(packets, dtype=np.float64)
    src = remove_invalid_values(packets[0:step])
    transform = Transform.identity()
    transforms, inverses, path = [], [], []
    index = 0
    while index + step <= packets.shape[0]:
        dst = remove_invalid_values(packets[index:index + step])
        index += step
        icp = icp_factory(dst)
        transform = icp.estimate(src, transform, max_iter)
        inv = transform.inverse()
        transforms.append(transform)
        inverses.append(inv)
        path.append(inv.t.copy())
    return transforms, inverses, np.array(path).reshape(-1, 2)


def run_scan_to_map(packets, step=PACKETS_PER_FRAME, max_iter=20, icp_factory=None, max_frames=None):
    """EXTENSION, not in the reference (BASELINE.json configs[4]): the
    scan3d frames registered against a map that grows.  The map starts as frame 0 (filtered as
    examples/scan3d.rs:63-69 does); every later frame is registered against the whole map with
    the reference's own estimator (Icp3d::estimate, warm-started with the previous pose as
    scan3d.rs:131 does) and then appended at its registered pose.  The pose maps the scan into
    the map frame, so the path is its translation directly (no inverse()).
    `icp_factory(dst)` must return an object with .estimate and .append(points, transform).
    Returns (transforms, path_xy, map_handle)."""
    icp_factory = icp_factory or Icp3d
    packets = np.asarray(packets, dtype=np.float64)
    world = icp_factory(remove_invalid_values(packets[0:step]))
    transform = Transform.identity()
    transforms, path = [], []
    index = step
    while index + step <= packets.shape[0] and (max_frames is None or len(transforms) < max_frames):
        scan = remove_invalid_values(packets[index:index + step])
        index += step
        transform = world.estimate(scan, transform, max_iter)
        world.append(scan, transform)
        transforms.append(transform)
        path.append(transform.t.copy())
    return transforms, np.array(path).reshape(-1, 2), world


def main(argv=None):
    """Headless stand-ins for the reference's two example binaries: the trajectory they draw,
    printed.  `scan2d DIR` reads DIR/001.txt, 002.txt, ... (examples/scan2d.rs); `scan3d` runs the
    synthetic packet stream in the scans.hdf5 layout (the file itself is absent from the reference
    mount); `scan2map` is the growing-map extension on the same stream."""
    import argparse

    from . import synth

    ap = argparse.ArgumentParser(prog="python -m icp_rust_amd.harness", description=main.__doc__)
    ap.add_argument("loop", choices=["scan2d", "scan3d", "scan2map"])
    ap.add_argument("scan_dir", nargs="?", help="scan2d: directory with 001.txt, 002.txt, ...")
    ap.add_argument("--frames", type=int, default=8, help="scan3d / scan2map: synthetic frames")
    ap.add_argument("--max-iter", type=int, default=20)
    args = ap.parse_args(argv)
    if args.loop == "scan2d":
        if not args.scan_dir:
            ap.error("scan2d needs the scan directory")
        _, _, path = run_scan2d(args.scan_dir, max_iter=args.max_iter)
    elif args.loop == "scan3d":
        _, _, path = run_scan3d(synth.synthetic_scan3d_packets(PACKETS_PER_FRAME * args.frames), max_iter=args.max_iter)
    else:
        _, path, world = run_scan_to_map(synth.synthetic_scan3d_packets(PACKETS_PER_FRAME * (args.frames + 1)),
                                         max_iter=args.max_iter)
        print(f"# map: {world.target_count} points")
    for k, (x, y) in enumerate(path):
        print(f"{k:4d} {x:+.9f} {y:+.9f}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
