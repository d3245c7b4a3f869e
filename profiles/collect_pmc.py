"""Per-kernel sums of whatever counters a rocprofv3 --pmc pass collected (averaged over launches).

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES ... --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py ...
    python3 profiles/collect_pmc.py gpurun_out/pmc_sq [kernel-substring] > profiles/r02_..._pmc.txt
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("icp::", "")
            if want and want not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
    for k in sorted(acc):
        n = max(len(launches[k]), 1)
        print(f"{k}  ({n} launches; per-launch averages, summed over XCDs)")
        for c, v in sorted(acc[k].items()):
            print(f"    {c:28s} {v / n:16.1f}")


if __name__ == "__main__":
    main()
