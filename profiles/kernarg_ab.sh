#!/bin/bash
# does the placement of kernel arguments (HIP_FORCE_DEV_KERNARG: device memory instead of host memory) change the step?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--steps 400 --warmup 20 --converging-calls 0 --rotating-calls 0 --brute-steps 0 --cpu-iters 0 --gn-points 0 --nn-points 0"
for R in 1 2 3; do for K in unset 0 1; do
  if [ $K = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$K; fi
  python3 bench.py $B 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HIP_FORCE_DEV_KERNARG=$K', 'ms_per_step %.4f' % d['ms_per_step'])"
done; done
unset HIP_FORCE_DEV_KERNARG; python3 profiles/frame_trace.py run | cut -c1-70
HIP_FORCE_DEV_KERNARG=1 python3 profiles/frame_trace.py run | cut -c1-70
HIP_FORCE_DEV_KERNARG=0 python3 profiles/frame_trace.py run | cut -c1-70
