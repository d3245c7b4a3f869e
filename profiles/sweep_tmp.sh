cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_full_config.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for cfg in "1 0.45" "2 0.45" "2 0.3" "2 0.6" "2 0.8" "3 0.45" "3 0.7" "4 0.6"; do set -- $cfg; ICP_GRID2=$1 ICP_GRID2_R=$2 timeout -k 10 120 python3 profiles/tile_probe.py 2>&1 | grep "ms/step" | sed "s/^/grid2=$1 r=$2 /"; done
