"""One estimate(20) of a 28k-point frame (examples/scan3d.rs:113-133) under rocprofv3 --kernel-trace: the launches of the
call in order, with the gaps between them.  usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3
profiles/frame_trace.py run ; python3 profiles/frame_trace.py show DIR"""
import csv
import glob
import sys
import time

sys.path.insert(0, "/root/repo")


def run():
    import icp_rust_amd as I
    from icp_rust_amd import synth
    pk = synth.synthetic_scan3d_packets(150)
    s3 = synth.remove_invalid_values(pk[:75])
    d3 = synth.remove_invalid_values(pk[75:150])
    icp = I.Icp3d(d3)
    for _ in range(5):
        icp.estimate(s3, I.Transform(), 20)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        T, inner = icp.estimate(s3, I.Transform(), 20, return_info="inner")
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"frame {len(s3)} x {len(d3)}: estimate(20) median {1e3 * ts[len(ts) // 2]:.3f} ms min {1e3 * ts[0]:.3f} ms inner {list(inner)}")


def run_fresh():
    """a fresh handle per frame, warm-started pose: what the scan3d loop does"""
    import icp_rust_amd as I
    from icp_rust_amd import synth
    from icp_rust_amd.harness import remove_invalid_values
    pk = synth.synthetic_scan3d_packets(75 * 10)
    src = remove_invalid_values(pk[:75])
    T = I.Transform.identity()
    for k in range(9):
        icp = I.Icp3d(remove_invalid_values(pk[75 * k:75 * (k + 1)]))
        t0 = time.perf_counter()
        T = icp.estimate(src, T, 20)
        dt = time.perf_counter() - t0
        icp.close()
    print(f"last frame of a fresh-handle loop: estimate(20) {1e3 * dt:.3f} ms")


def show(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last call: from the last k_query / first kernel after a long gap
    starts = [int(r["Start_Timestamp"]) for r in rows]
    first = [i for i, r in enumerate(rows) if "k_nn_grid_seed" in r["Kernel_Name"] or "true, true, 4>" in r["Kernel_Name"]]
    cut = first[-2] if len(first) > 1 else first[-1]  # (the last call but one: complete for certain)
    rows = rows[:first[-1]] if len(first) > 1 else rows
    t0, prev_end = starts[cut], None
    for r in rows[cut:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = "" if prev_end is None else f" gap {(s - prev_end) / 1e3:6.1f}"
        print(f"{r['Kernel_Name'].split('(')[0].replace('void ', '')[:50]:50s} start {(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f}{gap}")
        prev_end = e


if __name__ == "__main__":
    {"run": run, "fresh": run_fresh}[sys.argv[1]]() if sys.argv[1] != "show" else show(sys.argv[2])
