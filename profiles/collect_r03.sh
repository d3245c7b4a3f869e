#!/bin/bash
# Everything under profiles/r03_* comes from this script, run on the MI355X box from the repo root:
#   gpurun -- 'bash profiles/collect_r03.sh'
# (counter passes are separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O
B="--brute-steps 0 --cpu-iters 0 --gn-points 0 --converging-calls 0"
echo "== default bench"; python3 bench.py > $O/bench_default_1M.json 2> $O/bench_default.err
echo "== kernel stats (grid)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py $B > $O/bench_grid_1M_under_rocprof.json 2> /dev/null
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/bench_grid_1M_kernel_stats.csv; rm -rf $O/kt
echo "== traffic PMC"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
python3 profiles/collect_traffic.py $O/pf $O/pw $O/traffic_pmc.json > $O/traffic_pmc.txt; rm -rf $O/pf $O/pw
echo "== TA PMC (search kernels)"
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $O/pt -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
python3 profiles/collect_pmc.py $O/pt k_nn_grid > $O/nn_grid_ta_pmc.txt; rm -rf $O/pt
if [ "${1:-}" = "full" ]; then
echo "== sweep: kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kb -- python3 bench.py --nn brute --steps 3 --warmup 1 --cpu-iters 0 --gn-points 0 > $O/bench_brute_1M_under_rocprof.json 2> /dev/null
cp $(find $O/kb -name '*kernel_stats.csv' | head -1) $O/bench_brute_1M_kernel_stats.csv; rm -rf $O/kb
echo "== gn_large: kernel stats + traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kg -- python3 profiles/gn_large_only.py > $O/gn_large_64M_under_rocprof.json 2> /dev/null
cp $(find $O/kg -name '*kernel_stats.csv' | head -1) $O/gn_large_64M_kernel_stats.csv; rm -rf $O/kg
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/gf -- python3 profiles/gn_large_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/gw -- python3 profiles/gn_large_only.py > /dev/null 2>&1
python3 profiles/collect_traffic.py $O/gf $O/gw $O/traffic_pmc_gn_large.json > $O/traffic_pmc_gn_large.txt; rm -rf $O/gf $O/gw
echo "== reference-sized inputs"; python3 bench_small.py > $O/bench_small.txt 2>&1
echo "== two ranks sharing this GPU over gloo (functional rehearsal of bench --gpus 2; not a measurement)"
ICP_BENCH_SHARE_GPU=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 40 --warmup 2 --brute-steps 0 --weak-steps 20 2> $O/bench_2rank.err | grep '^{' > $O/bench_2rank_shared_gloo.json
echo "== virtual ranks: cost of the exchanges"; python3 profiles/multi_virtual_timing.py > $O/multi_virtual_timing.txt 2>&1
echo "== map"; python3 bench_map.py > $O/bench_map_10M.json 2> /dev/null
echo "== timelines: steady state of the 1M pair, one 28k-point frame"
rocprofv3 --kernel-trace --output-format csv -d $O/ks -- python3 bench.py $B --steps 60 > /dev/null 2>&1
python3 profiles/steady_state_timeline.py $O/ks > $O/timeline_steady_state.txt; rm -rf $O/ks
rocprofv3 --kernel-trace --output-format csv -d $O/kf -- python3 profiles/frame28k_trace.py > $O/frame28k_run.txt 2>&1
TAIL=75 python3 profiles/frame28k_trace.py --analyze $O/kf > $O/frame28k_timeline.txt; rm -rf $O/kf
echo "== evaluation pipelines side by side (profiles/ab_libs.py: fused everywhere / default / search stream only / never)"
python3 profiles/ab_libs.py 2 icp_rust_amd/lib/libicp_mi355x.so:ICP_WIN_FUSE_MODE=1 icp_rust_amd/lib/libicp_mi355x.so icp_rust_amd/lib/libicp_mi355x.so:ICP_WIN_FUSE_MODE=4 icp_rust_amd/lib/libicp_mi355x.so:ICP_WIN_NO_FUSE=1 > $O/eval_fusion_ab.txt 2>&1
echo "== matches that survive an outer iteration, certificates"; python3 profiles/nn_stability.py > $O/nn_stability.txt 2>&1
python3 profiles/settled_phase.py > $O/settled_phase.txt 2>&1; ICP_NN_NO_CERT=1 python3 profiles/settled_phase.py >> $O/settled_phase.txt 2>&1
fi
ls -la $O
