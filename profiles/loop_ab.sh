#!/bin/bash
# same-box A/B of library variants on the workloads the one-launch loops serve: converging pair (k_gn_loop), virtual ranks
# (k_gn_loop_shard), interleaved repeats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for R in 1 2 3; do for V in "$@"; do
  if [ "$V" = default ]; then unset ICP_MI355X_LIB; else export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_ab_$V.so; fi
  echo "$V: $(python3 profiles/loop_probe.py conv 2>&1 | grep '^converging' | cut -c1-40)"
done; done
for V in "$@"; do
  if [ "$V" = default ]; then unset ICP_MI355X_LIB; else export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_ab_$V.so; fi
  echo "== $V"; python3 profiles/multi_virtual_timing.py 2>&1 | grep "virtual ranks" | cut -c1-90
done
