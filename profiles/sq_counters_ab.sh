#!/bin/bash
# SQ instruction counters of the search kernel for a list of library builds (bench.py's headline leg under rocprofv3 --pmc):
#   bash profiles/sq_counters_ab.sh mi355x ab_m_x0r4 ...     (names after icp_rust_amd/lib/libicp_)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--brute-steps 0 --cpu-iters 0 --gn-points 0 --converging-calls 0 --rotating-calls 0"
for v in "$@"; do
  export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_$v.so
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/sq_$v -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
  echo "== $v"; python3 profiles/collect_pmc.py gpurun_out/sq_$v k_nn_grid_warm; rm -rf gpurun_out/sq_$v
done
