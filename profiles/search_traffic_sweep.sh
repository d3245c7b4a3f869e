#!/bin/bash
# Fabric-side bytes fetched by the warm search per launch for several XCD chunk sizes (experiments build):
#   gpurun -- 'bash profiles/search_traffic_sweep.sh'   ->  gpurun_out/search_traffic_sweep.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sts; mkdir -p $O
export SEARCH_PROBE_CHILD=1 ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_mi355x_exp.so
for C in 0 8 16 32 64 128; do
  export ICP_NN_XCD_CHUNK=$C
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f$C -- python3 profiles/search_probe.py > $O/run_$C.txt 2> $O/err_$C.txt
  python3 - $O/f$C $C <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d, C = sys.argv[1], sys.argv[2]
per = defaultdict(float); names = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE" or "k_nn_grid_warm" not in r["Kernel_Name"]: continue
        per[r.get("Dispatch_Id") or r.get("Correlation_Id")] += float(r["Counter_Value"])
v = sorted(per.values())
print(f"chunk {C:>3}: warm search FETCH_SIZE {sum(v) / max(len(v), 1) / 1024:.1f} MB per launch ({len(v)} launches), "
      + (open(os.path.join(os.path.dirname(d), f"run_{C}.txt")).read().strip().splitlines() or ["(no output)"])[-1])
PY
  rm -rf $O/f$C
done
