"""Timeline of one reference-sized registration (BASELINE configs[1] stand-in: a 28.8k-point 3-D
window, estimate(src, I, 20)) for rocprofv3 --kernel-trace: where the ~2 ms go.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt28 -- python3 profiles/frame28k_trace.py
    python3 profiles/frame28k_trace.py --analyze gpurun_out/kt28
"""
import csv, glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run2d():
    import icp_rust_amd as I
    from icp_rust_amd.scans import load_scan2d
    G = os.path.join(ROOT, "tests", "golden", "scans2d")
    src = load_scan2d(f"{G}/001.txt"); dst = load_scan2d(f"{G}/002.txt")
    icp = I.Icp2d(dst)
    for _ in range(3):
        icp.estimate(src, I.Transform(), 20)
    t0 = time.perf_counter()
    for _ in range(10):
        inner = icp.estimate(src, I.Transform(), 20, return_info=True)[-1]
    print(f"2-D scan: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms per estimate(20), inner {inner.tolist()}")


def run():
    import numpy as np
    import icp_rust_amd as I
    from icp_rust_amd import synth
    pk = synth.synthetic_scan3d_packets(150)
    s3 = synth.remove_invalid_values(pk[:75]); d3 = synth.remove_invalid_values(pk[75:150])
    icp = I.Icp3d(d3)
    for _ in range(3):
        icp.estimate(s3, I.Transform(), 20)
    t0 = time.perf_counter()
    for _ in range(10):
        inner = icp.estimate(s3, I.Transform(), 20, return_info=True)[-1]
    print(f"28k frame: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms per estimate(20), inner {inner.tolist()}")


def analyze(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("icp::", "")))
    rows.sort()
    # the last estimate call: from the last k_query_count on
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_query_count")]
    if starts:
        seg = rows[starts[-1]:]
    else:  # small clouds take no snapshot: show the tail of the trace
        seg = rows[-int(os.environ.get("TAIL", "260")):]
    t0 = seg[0][0]
    busy = 0
    prev_end = t0
    per = {}
    for s, e, n in seg:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {n}")
        prev_end = max(prev_end, e)
        per.setdefault(n, [0, 0.0]); per[n][0] += 1; per[n][1] += (e - s) / 1e3
    print("span us", (prev_end - t0) / 1e3)
    for n, (c, t) in sorted(per.items(), key=lambda x: -x[1][1]):
        print(f"{n:50s} {c:4d} {t:9.1f} us  avg {t / c:6.1f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
        analyze(sys.argv[2])
    elif len(sys.argv) > 1 and sys.argv[1] == "--scan2d":
        run2d()
    else:
        run()
