cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/m1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 profiles/multi_one.py 1 > $O/out.txt 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/kt
grep "W=" $O/out.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/m1/kernel_stats.csv')))
for r in rows[:14]:
    print("%-60s n=%5s avg=%9.1f us  %5.1f%%"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
