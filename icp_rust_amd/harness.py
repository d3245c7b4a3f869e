"""Headless frame loops of the reference's two examples: same file
handling, same warm start, same per-frame `Icp::new` + `estimate(src, transform, 20)`, the
trajectory instead of a window.

`icp_factory(dst) -> object with .estimate(src, Transform, max_iter) -> Transform` defaults to
the GPU classes; tests inject an oracle-backed factory to check the loop semantics on CPU.
"""
import os

import numpy as np

from .api import Icp2d, Icp3d, Transform
from .scans import PacketFile, load_scan2d, write_packets
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


class _ArrayPackets:
    """`Scan` (examples/scan3d.rs:17-61) over an in-memory packet array (n_packets, 384, 3)."""

    def __init__(self, packets):
        self.p = np.asarray(packets, dtype=np.float64)

    def size(self):
        return self.p.shape[0]

    def get_range(self, start, end):
        return self.p[start:end].reshape(-1, 3)


def _packet_source(packets):
    return packets if hasattr(packets, "get_range") and hasattr(packets, "size") else _ArrayPackets(packets)


def run_scan3d(packets, step=PACKETS_PER_FRAME, max_iter=20, icp_factory=None, pipeline=True, timings=None):
    """examples/scan3d.rs:104-158 on a packet stream -- an in-memory array (n_packets, 384, 3) or a
    `scans.PacketFile` (the reference's `Scan` over scans.hdf5): the source is packets [0, step) with
    invalid returns removed (:104-107, 63-69); every frame takes dst = packets [index, index+step),
    filtered, THEN advances index (:113-121) -- so the first frame registers the source against itself;
    Icp3d::new(&dst); estimate warm-started (:130-131); path of transform.inverse().t (:133,144).

    `pipeline`: frame k+1's packets are read and filtered, uploaded and its
    search structure built (`Icp3d::new`) by a second host thread on that handle's own stream WHILE frame
    k estimates -- a ring of two handles; the registration itself is untouched, so the trajectory is the
    same bit for bit.  `timings` (optional list) receives the wall time of every frame in seconds.
    Returns (transforms, inverse_transforms, path_xy)."""
    import time
    from concurrent.futures import ThreadPoolExecutor

    icp_factory = icp_factory or Icp3d
    scan = _packet_source(packets)
    src = remove_invalid_values(scan.get_range(0, step))
    transform = Transform.identity()
    transforms, inverses, path = [], [], []

    def make(index):  # what the frame loop does before estimate: Scan::get_range + filter + Icp3d::new
        return icp_factory(remove_invalid_values(scan.get_range(index, index + step)))

    index = 0
    pool = ThreadPoolExecutor(max_workers=1) if pipeline else None
    ahead = pool.submit(make, index) if pool and index + step <= scan.size() else None
    try:
        while index + step <= scan.size():
            t0 = time.perf_counter()
            icp = ahead.result() if ahead is not None else make(index)
            ahead = None
            index += step
            ahead = pool.submit(make, index) if pool and index + step <= scan.size() else None
            transform = icp.estimate(src, transform, max_iter)
            if hasattr(icp, "close"):
                icp.close()  # back to the handle pool: the NEXT make() reuses its buffers and streams
            inv = transform.inverse()
            transforms.append(transform)
            inverses.append(inv)
            path.append(inv.t.copy())
            if timings is not None:
                timings.append(time.perf_counter() - t0)
    finally:
        if pool:
            pool.shutdown(wait=True)
        # (an error above can leave the prefetched handle of the next frame unconsumed)
        if ahead is not None and ahead.done() and ahead.exception() is None and hasattr(ahead.result(), "close"):
            ahead.result().close()
    return transforms, inverses, np.array(path).reshape(-1, 2)


def run_scan_to_map(packets, step=PACKETS_PER_FRAME, max_iter=20, icp_factory=None, max_frames=None,
                    point_to_plane=None):
    """EXTENSION, not in the reference (BASELINE.json configs[4]): the
    scan3d frames registered against a map that grows.  The map starts as frame 0 (filtered as
    examples/scan3d.rs:63-69 does); every later frame is registered against the whole map with
    the reference's own estimator (Icp3d::estimate, warm-started with the previous pose as
    scan3d.rs:131 does) and then appended at its registered pose.  The pose maps the scan into
    the map frame, so the path is its translation directly (no inverse()).
    `icp_factory(dst)` must return an object with .estimate and .append(points, transform).
    `point_to_plane=k` (BASELINE configs[4] names point-to-plane): the frames are registered with the
    point-to-plane residual instead (Icp3d.estimate_point_to_plane, the other labelled extension); every
    map point carries the normal of its k nearest map points AT THE TIME IT WAS INSERTED
    (compute_normals for frame 0, update_normals after every append).
    Returns (transforms, path_xy, map_handle)."""
    icp_factory = icp_factory or Icp3d
    packets = np.asarray(packets, dtype=np.float64)
    world = icp_factory(remove_invalid_values(packets[0:step]))
    if point_to_plane:
        world.compute_normals(point_to_plane)
    transform = Transform.identity()
    transforms, path = [], []
    index = step
    while index + step <= packets.shape[0] and (max_frames is None or len(transforms) < max_frames):
        scan = remove_invalid_values(packets[index:index + step])
        index += step
        if point_to_plane:
            transform = world.estimate_point_to_plane(scan, transform, max_iter)
        else:
            transform = world.estimate(scan, transform, max_iter)
        world.append(scan, transform)
        if point_to_plane:
            world.update_normals(point_to_plane)
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
    ap.add_argument("loop", choices=["scan2d", "scan3d", "scan2map", "write-synth"])
    ap.add_argument("scan_dir", nargs="?", help="scan2d: directory with 001.txt, 002.txt, ...; scan3d / scan2map / "
                                                "write-synth: packet container file")
    ap.add_argument("--frames", type=int, default=8, help="scan3d / scan2map: synthetic frames")
    ap.add_argument("--max-iter", type=int, default=20)
    ap.add_argument("--point-to-plane", type=int, default=0, metavar="K",
                    help="scan2map: register with point-to-plane residuals, normals from K nearest map points (extension)")
    args = ap.parse_args(argv)
    if args.loop == "scan2d":
        if not args.scan_dir:
            ap.error("scan2d needs the scan directory")
        _, _, path = run_scan2d(args.scan_dir, max_iter=args.max_iter)
    elif args.loop == "write-synth":
        if not args.scan_dir:
            ap.error("write-synth needs the output file")
        write_packets(args.scan_dir, synth.synthetic_scan3d_packets(PACKETS_PER_FRAME * args.frames))
        print(f"# wrote {PACKETS_PER_FRAME * args.frames} packets to {args.scan_dir}")
        return 0
    elif args.loop == "scan3d":
        stream = PacketFile(args.scan_dir) if args.scan_dir else synth.synthetic_scan3d_packets(PACKETS_PER_FRAME * args.frames)
        _, _, path = run_scan3d(stream, max_iter=args.max_iter)
    else:
        stream = PacketFile(args.scan_dir).as_array() if args.scan_dir else \
            synth.synthetic_scan3d_packets(PACKETS_PER_FRAME * (args.frames + 1))
        _, path, world = run_scan_to_map(stream, max_iter=args.max_iter, point_to_plane=args.point_to_plane or None)
        print(f"# map: {world.target_count} points")
    for k, (x, y) in enumerate(path):
        print(f"{k:4d} {x:+.9f} {y:+.9f}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
