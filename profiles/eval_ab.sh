#!/bin/bash
# stand-alone kernel durations of one evaluation, second pass over the points (ICP_WIN_BKT=0) against buckets (=1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_mi355x_exp.so
for B in 0 1; do
  export ICP_WIN_BKT=$B
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ke$B -- python3 profiles/eval_probe.py ${1:-1000000} > gpurun_out/eval_probe_$B.txt 2>&1
  echo "== ICP_WIN_BKT=$B"; tail -1 gpurun_out/eval_probe_$B.txt
  python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/ke$B/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_win' in r['Name'] or 'k_pull' in r['Name']:
        print(f"  {r['Name'].split('(')[0][:40]:40s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f} us min {float(r['MinNs'])/1e3:7.1f}")
PY
  rm -rf gpurun_out/ke$B
done
