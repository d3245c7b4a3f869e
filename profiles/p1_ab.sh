#!/bin/bash
# the headline bench + kernel averages of one build (default library, or the variant named in $1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$1" ] && export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_ab_$1.so
python3 bench.py --converging-calls 3 --rotating-calls 0 > gpurun_out/b1.json 2> gpurun_out/b1.err; python3 -c "
import json; d=json.loads(open('gpurun_out/b1.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'it/s', d['value'], 'search ms', d['roofline']['avg_launch_ms'], 'converging', d.get('converging',{}).get('ms_per_step'))"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k1 -- python3 bench.py --steps 200 --warmup 20 --converging-calls 0 --rotating-calls 0 --brute-steps 0 > gpurun_out/b1p.txt 2>&1
python3 profiles/stats_top.py gpurun_out/k1 8 2>/dev/null
rm -rf gpurun_out/k1
