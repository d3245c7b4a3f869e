cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/mapk; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench_map.py --frames 10 > $O/out.txt 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/kt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/mapk/kernel_stats.csv')))
for r in rows[:18]:
    print("%-60s n=%5s avg=%9.1f us tot=%8.2f ms"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
