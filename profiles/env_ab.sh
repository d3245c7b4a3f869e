#!/bin/bash
# same-box A/B of the headline pair over environment settings of the experiments build:  env_ab.sh "A=1" "B=2 C=3" ...
cd "$GRAFT_REPO_ROOT"
export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_mi355x_exp.so
for round in 1 2 3; do
  for E in "$@"; do
    echo -n "$E: "; env $E python3 ${PROBE:-profiles/ahead_probe.py} 2>&1 | grep -E "estimate|ms"
  done
done
