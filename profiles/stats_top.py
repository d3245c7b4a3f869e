"""Top kernels of a rocprofv3 --kernel-trace --stats run: python3 profiles/stats_top.py DIR [rows] [name filter]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10
flt = sys.argv[3] if len(sys.argv) > 3 else ""
for r in [r for r in csv.DictReader(open(f)) if flt in r["Name"]][:rows]:
    name = r["Name"].split("(")[0].replace("void ", "")[:56]
    print(f"  {name:56s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us min {float(r['MinNs']) / 1e3:9.1f} total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
