cd $GRAFT_REPO_ROOT
for v in "X=1" "ICP_WIN_ALL_CORESIDENT=1"; do
  env $v python3 bench_small.py 2>&1 | grep "3D scan\|frame loop (12 frames of ~28k points, frame" | cut -c1-120 | sed "s/^/[$v] /"
  env $v python3 bench.py --steps 40 --brute-steps 0 --cpu-iters 0 --gn-points 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], d['converging_pair']['ms_per_step'])"
done
