"""Every kernel of the LAST estimate(20) call of profiles/call_head_trace.py's run, in start order, with the idle time of
the device in front of it (start minus the latest end so far): where the host, not a kernel, is on the critical path.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kh -- python3 profiles/call_head_trace.py run
    python3 profiles/call_gaps.py gpurun_out/kh [marker of a call's first kernel]"""
import csv, glob, os, re, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("icp::", "")[:40]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_query_cell"  # (the first kernel of a call; a 28k frame: "true, true, 4>")
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
i0 = starts[-2] if len(starts) > 1 else starts[-1]  # (the last call but one: complete for certain)
rows = rows[:starts[-1]] if len(starts) > 1 else rows
t0 = int(rows[i0]["Start_Timestamp"])
last_end = t0
idle = 0.0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end) / 1e3
    if gap > 0: idle += gap
    print(f"{name(r):40s} start {(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:6.1f} idle before {gap:6.1f}" + ("   <--" if gap > 1.5 else ""))
    last_end = max(last_end, e)
print(f"call: {(last_end - t0) / 1e3:.1f} us from its first launch to its last kernel's end, device idle {idle:.1f} us of it")
