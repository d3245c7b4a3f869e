#!/bin/bash
# estimate(20) on the 1M pair under experiment knobs (experiments build): bash profiles/ahead_ab.sh "" ICP_WIN_FUSE_MODE=1 ...
export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_mi355x_exp.so
for rnd in 1 2; do
for kv in "$@"; do
  echo -n "${kv:-(default)}: "
  env $kv ICP_STEP_TRACE=1 timeout -k 10 200 python3 profiles/ahead_probe.py 2>&1 | grep -E "estimate\(20\)|step trace" | tail -n 2 | tr '\n' ' '; echo
done; done
