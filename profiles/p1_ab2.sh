#!/bin/bash
# same-box A/B of library variants: stand-alone evaluation kernels, then the headline step (3 repeats each, interleaved)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for V in "$@"; do
  if [ "$V" = default ]; then unset ICP_MI355X_LIB; else export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_ab_$V.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kk -- python3 profiles/eval_probe.py 1000000 300 > gpurun_out/kk.txt 2>&1
  echo "== $V alone"; python3 profiles/stats_top.py gpurun_out/kk 2 k_win
  rm -rf gpurun_out/kk
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kk -- python3 bench.py --steps 200 --warmup 20 --converging-calls 0 --rotating-calls 0 --brute-steps 0 --cpu-iters 0 > gpurun_out/kk.txt 2>&1
  echo "== $V in the step"; python3 profiles/stats_top.py gpurun_out/kk 3 k_
  rm -rf gpurun_out/kk
done
for R in 1 2 3; do for V in "$@"; do
  if [ "$V" = default ]; then unset ICP_MI355X_LIB; else export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_ab_$V.so; fi
  python3 bench.py --steps 400 --warmup 20 --converging-calls 0 --rotating-calls 0 --brute-steps 0 --cpu-iters 0 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', 'ms_per_step %.4f' % d['ms_per_step'])"
done; done
