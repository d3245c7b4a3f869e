#!/bin/bash
# stand-alone duration of k_win_hist_sums_bkt for library variants (wrong results on purpose: what bounds the launch?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for V in "$@"; do
  if [ "$V" = default ]; then unset ICP_MI355X_LIB; else export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_ab_$V.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kk -- python3 profiles/eval_probe.py 1000000 300 > gpurun_out/kk.txt 2>&1
  echo "== $V"; python3 profiles/stats_top.py gpurun_out/kk 3 k_win
  rm -rf gpurun_out/kk
done
