#!/bin/bash
# round-3 development probe: per-kernel times of the LDS-tile search on the benchmark pair, and a sweep of
# the grid geometry (gpurun -- 'bash profiles/run_probe_r03.sh [sweep]')
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/p1
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 profiles/tile_probe.py > $O/out.txt 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/kt
grep -v amdgpu.ids $O/out.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/p1/kernel_stats.csv')))
for r in rows[:12]:
    print("%-60s n=%5s avg=%9.1f us  %5.1f%%"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
if [ "$1" = "sweep" ]; then
  for cfg in "4 4" "8 4" "8 8" "16 8"; do set -- $cfg; ICP_GRID_OCC=$1 ICP_GRID_FX=$2 timeout -k 10 120 python3 profiles/tile_probe.py 2>&1 | grep -v amdgpu.ids; done
  ICP_TILE_NO_XCD=1 timeout -k 10 120 python3 profiles/tile_probe.py 2>&1 | grep -v amdgpu.ids
  ICP_NN_NO_TILE=1 timeout -k 10 120 python3 profiles/tile_probe.py 2>&1 | grep -v amdgpu.ids
fi
if [ "$1" = "phases" ]; then
  ICP_MI355X_LIB=icp_rust_amd/lib/libicp_mi355x_stats.so python3 profiles/tile_phases.py 2>&1 | grep -v amdgpu.ids
  ICP_QSORT_BLOCK=0 timeout -k 10 120 python3 profiles/tile_probe.py 2>&1 | grep -v amdgpu.ids
  ICP_QSORT_BLOCK=2 timeout -k 10 120 python3 profiles/tile_probe.py 2>&1 | grep -v amdgpu.ids
  ICP_GRID_OCC=8 ICP_QSORT_BLOCK=1 timeout -k 10 120 python3 profiles/tile_probe.py 2>&1 | grep -v amdgpu.ids
  ICP_GRID_OCC=8 ICP_GRID_FX=8 ICP_QSORT_BLOCK=1 timeout -k 10 120 python3 profiles/tile_probe.py 2>&1 | grep -v amdgpu.ids
  ICP_GRID_OCC=4 ICP_GRID_FX=4 ICP_QSORT_BLOCK=2 timeout -k 10 120 python3 profiles/tile_probe.py 2>&1 | grep -v amdgpu.ids
  ICP_TILE_NO_XCD=1 timeout -k 10 120 python3 profiles/tile_probe.py 2>&1 | grep -v amdgpu.ids
fi
