#!/bin/bash
# same-box A/B of the headline pair: second pass over the points (ICP_WIN_BKT=0) against filed candidates (1), experiments build
cd "$GRAFT_REPO_ROOT"
export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_mi355x_exp.so
for round in 1 2 3; do
  for B in ${@:-0 1}; do
    echo -n "ICP_WIN_BKT=$B: "; ICP_WIN_BKT=$B python3 profiles/ahead_probe.py 2>&1 | grep estimate
  done
done
