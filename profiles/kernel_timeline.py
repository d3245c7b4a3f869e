"""Consecutive kernels from a rocprofv3 kernel trace, the last N of them (or a window):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 <script>
    python3 profiles/kernel_timeline.py gpurun_out/kt [count] [skip_from_end]
"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                     r.get("Queue_Id", "?")))
rows.sort()
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sel = rows[len(rows) - count - skip:len(rows) - skip]
t0 = sel[0][0]
qs = {}
for s, e, n, q in sel:
    qn = qs.setdefault(q, f"q{len(qs) + 1}")
    print(f"{n[:48]:48s} {qn} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:6.1f}")
