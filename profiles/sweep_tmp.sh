cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_full_config.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 || exit 1
bash profiles/run_probe_r03.sh 2>&1 | grep -v "^W2026\|^E2026"
ICP_MI355X_LIB=icp_rust_amd/lib/libicp_mi355x_stats.so python3 profiles/tile_phases.py 2>&1 | grep -v amdgpu.ids
for b in 1 2; do for o in 2 4 8; do ICP_QSORT_BLOCK=$b ICP_GRID_OCC=$o timeout -k 10 120 python3 profiles/tile_probe.py 2>&1 | grep "ms/step\|fallback" | sed "s/^/blk=$b /"; done; done
