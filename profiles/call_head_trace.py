"""The head of one estimate(20) call on the 1M pair: every kernel from the call's first launch to its third search,
with start / end relative to the first (run under rocprofv3 --kernel-trace; second argument: the trace directory).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kh -- python3 profiles/call_head_trace.py run
    python3 profiles/call_head_trace.py analyze gpurun_out/kh"""
import csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import torch, icp_rust_amd as I
    from icp_rust_amd import synth
    src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    icp = I.Icp3d(d_dst)
    for _ in range(4):
        icp.estimate(d_src, I.Transform(), 20)
    torch.cuda.synchronize()
else:
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("icp::", "")[:46]
    seeded = [i for i, r in enumerate(rows) if "k_nn_grid_seeded" in r["Kernel_Name"] or "k_nn_grid_seed<" in r["Kernel_Name"]]
    i0 = seeded[-1]
    # back up to the first kernel of that call: the query-cell kernel
    while i0 > 0 and "k_nn_grid_warm_coop" not in rows[i0 - 1]["Kernel_Name"] and "k_unpermute" not in rows[i0 - 1]["Kernel_Name"] and "k_win" not in rows[i0 - 1]["Kernel_Name"]:
        i0 -= 1
    t0 = int(rows[i0]["Start_Timestamp"])
    nsearch = 0
    for r in rows[i0:]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{name(r):46s} start {s:8.1f} end {e:8.1f} dur {e - s:7.1f} us")
        if "k_nn_grid_warm_coop" in r["Kernel_Name"]:
            nsearch += 1
            if nsearch == 3: break
