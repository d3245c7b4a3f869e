"""Kernel timeline of the pipelined scan3d frame loop: frame k+1's Icp3d::new
(upload, bounding box, grid build) on its own stream while frame k estimates.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 profiles/scan3d_timeline.py
    python3 profiles/scan3d_timeline.py --analyze DIR
"""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BUILD = ("k_grid_bbox", "k_grid_count", "k_scan_local", "k_scan_totals", "k_scan_add", "k_grid_scatter")


def run():
    from icp_rust_amd import harness, synth

    pk = synth.synthetic_scan3d_packets(75 * 10)
    harness.run_scan3d(pk[:75 * 3])
    for piped in (False, True):
        tm = []
        harness.run_scan3d(pk, pipeline=piped, timings=tm)
        print(f"pipeline={piped}: frames after the first two {1e3 * sum(tm[2:]) / len(tm[2:]):.3f} ms each")


def analyze(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("icp::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
    rows.sort()
    # the last (pipelined) run: take the last 3 grid builds and everything between the first and the last of them
    builds = [i for i, r in enumerate(rows) if r[2].startswith("k_grid_bbox")]
    lo, hi = builds[-4], builds[-1]
    seg = rows[lo:hi + 8]
    t0 = seg[0][0]
    print("start_us   end_us    dur_us  queue  kernel      (B = part of the NEXT frame's Icp3d::new)")
    overl = 0
    est_busy = []
    for s, e, n, q in seg:
        is_build = n.startswith(BUILD)
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {q:>5}  {'B ' if is_build else '  '}{n}")
        if not is_build:
            est_busy.append((s, e))
    for s, e, n, q in seg:
        if n.startswith(BUILD):
            overl += any(s < ee and e > ss for ss, ee in est_busy)
    nb = sum(1 for r in seg if r[2].startswith(BUILD))
    print(f"build kernels in the window: {nb}, of which overlapping an estimate kernel in time: {overl}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
        analyze(sys.argv[2])
    else:
        run()
