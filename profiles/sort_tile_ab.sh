#!/bin/bash
# call head of the 1M pair under rocprofv3 for a list of library builds (names after icp_rust_amd/lib/libicp_), one box:
#   bash profiles/sort_tile_ab.sh mi355x ab_rs512 ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_$v.so
  python3 -m pytest tests/test_gpu_sort.py -x -q 2>&1 | tail -1
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kh -- python3 profiles/call_head_trace.py run > /dev/null 2>&1
  echo "== $v"; python3 profiles/call_head_trace.py analyze gpurun_out/kh > gpurun_out/kh.txt; sed -n 1,8p gpurun_out/kh.txt; rm -rf gpurun_out/kh gpurun_out/kh.txt
done
