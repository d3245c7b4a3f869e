"""What a sharded outer iteration enqueues, from a rocprofv3 kernel trace of icp_multi_estimate (VERDICT r3 item 1:
"<= 6 enqueues and 0 host waits per rank per evaluation shown in a kernel trace").

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 profiles/multi_trace_count.py run W
    python3 profiles/multi_trace_count.py analyze DIR W

`run`: W virtual ranks on the 1M benchmark pair: two warm-up calls, a marker, then ONE estimate(20).  `analyze`: the
launches of that last call by kernel name; per outer iteration and rank: enqueues, of which inner-loop launches; the
evaluations they served come from the library's counters (printed by `run`)."""
import csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)

def run(W):
    import icp_rust_amd as I
    from icp_rust_amd import synth
    src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
    mu = I.IcpMulti(dst, [0] * W)
    mu.estimate(src, I.Transform(), 20)
    mu.estimate(src, I.Transform(), 20)
    c0, l0 = mu.counters(), mu.loop_counters()
    T, _, inner = mu.estimate(src, I.Transform(), 20, return_info=True)
    c1, l1 = mu.counters(), mu.loop_counters()
    print(f"W={W}: last estimate(20): inner {inner.tolist()}; evaluations sharded {c1[0] - c0[0]}, replicated {c1[1] - c0[1]}; "
          f"inner-loop launches {l1[0] - l0[0]} (per rank), evaluations they served {l1[1] - l0[1]}, handed back {l1[2] - l0[2]}")
    mu.close()

def analyze(d, W):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("icp::", "")
    # the last call = everything after the last but 20*... simply: the last third of the searches (3 equal calls)
    loops = [i for i, r in enumerate(rows) if "k_gn_loop_shard" in r["Kernel_Name"]]
    per_call = len(loops) // 3
    first = loops[-per_call] if per_call else 0
    # back up to the search launches in front of the call's first loop launch
    while first > 0 and "k_nn_grid" in rows[first - 1]["Kernel_Name"]:
        first -= 1
    last = rows[first:]
    counts = {}
    for r in last:
        counts[name(r)] = counts.get(name(r), 0) + 1
    total = len(last)
    print(f"W={W}: launches of the last estimate(20) by kernel (per-call snapshot / sort / index kernels included):")
    for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
        print(f"  {v:5d}  {k}")
    nloop = counts.get("k_gn_loop_shard", 0)
    steady = {k: v for k, v in counts.items() if k.startswith("k_nn_grid_warm") or k.startswith("k_gn_loop_shard")}
    print(f"  total {total}; per outer iteration (20): {total / 20:.1f} launches for {W} rank(s) = {total / 20 / W:.2f} per rank; "
          f"steady-state kernels (warm search + inner-loop launch): {sum(steady.values()) / 20:.1f} per iteration; "
          f"inner-loop launches {nloop} ({'one launch carries all ranks of this device' if W > 1 else 'one per rank'})")
    t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
    print(f"  span {1e-6 * (t1 - t0):.3f} ms, kernels busy {1e-6 * busy:.3f} ms (sum over launches; ranks' kernels on one stream)")

if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        analyze(sys.argv[2], int(sys.argv[3]))
