"""Consecutive kernels of the steady state of bench.py's timed region from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --brute-steps 0 --cpu-iters 0 --gn-points 0
    python3 profiles/steady_state_timeline.py gpurun_out/kt > profiles/r02_timeline_steady_state.txt
"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                     r.get("Queue_Id", "?")))
rows.sort()
warm = [i for i, r in enumerate(rows) if r[2].startswith("icp::k_nn_grid<3, true, false") or r[2].startswith("icp::k_nn_grid_warm")]
start = warm[len(warm) * 2 // 3]  # well inside a 20-iteration call
t0 = rows[start][0]
print("rocprofv3 --kernel-trace of `python3 bench.py --brute-steps 0 --cpu-iters 0 --gn-points 0`: consecutive kernels of the "
      "steady state of the timed region (us).")
print("q = HW queue: one is the handle's stream (search + first evaluation of every outer iteration), the other the "
      "evaluation stream (the evaluation that decides the speculated pose).\n")
qs = {}
for s, e, n, q in rows[start:start + 30]:
    qn = qs.setdefault(q, f"q{len(qs) + 1}")
    print(f"{n[:44]:44s} {qn} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:6.1f}")
