#!/bin/bash
# average duration of the steady-state search kernel for a list of library builds (names after icp_rust_amd/lib/libicp_), one box:
#   bash profiles/search_kernel_ab.sh mi355x ab_heads mi355x_exp:ICP_GRID_FX=8 ...   (name[:ENV=value]...)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--brute-steps 0 --cpu-iters 0 --gn-points 0 --nn-points 0 --converging-calls 0 --rotating-calls 0"
for round in 1 2; do
for spec in "$@"; do
  v=${spec%%:*}
  export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_$v.so
  envs=""; if [ "$spec" != "$v" ]; then envs=$(echo "${spec#*:}" | tr ':' ' '); fi
  for kv in $envs; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --steps 20 --warmup 5 $B > gpurun_out/ks.json 2> /dev/null
  echo "== $spec: $(python3 -c "import json;d=json.loads(open('gpurun_out/ks.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms per step under the tracer"
  python3 - "$(find gpurun_out/ks -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("warm_coop", "seeded", "hist_sums_bkt2", "win_pick2")):
        print(f'   {r["Name"].split("(")[0][-40:]:42s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:7.2f} us min {float(r["MinNs"]) / 1e3:7.2f}')
PY
  rm -rf gpurun_out/ks gpurun_out/ks.json
  for kv in $envs; do unset "${kv%%=*}"; done
done
done
