#!/bin/bash
# Everything under profiles/r06_* comes from this script, run on the MI355X box from the repo root:
#   gpurun -- 'bash profiles/collect_r06.sh [full]'
# (counter passes are separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06
mkdir -p $O
B="--brute-steps 0 --cpu-iters 0 --gn-points 0 --nn-points 0 --converging-calls 0 --rotating-calls 0"
echo "== default bench (what the driver runs)"; python3 bench.py > $O/bench_default_1M.json 2> $O/bench_default.err
echo "== kernel stats (grid, headline pair)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 5 $B > $O/bench_grid_1M_under_rocprof.json 2> /dev/null
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/bench_grid_1M_kernel_stats.csv
python3 profiles/kernel_timeline.py $O/kt 24 30 > $O/timeline_steady_state.txt; rm -rf $O/kt
echo "== kernel stats (converging pair: the one-launch inner loops)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kc -- python3 profiles/loop_probe.py conv > $O/converging_run.txt 2> /dev/null
cp $(find $O/kc -name '*kernel_stats.csv' | head -1) $O/converging_kernel_stats.csv; rm -rf $O/kc
echo "== traffic PMC (headline)"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
python3 profiles/collect_traffic.py $O/pf $O/pw $O/traffic_pmc.json > $O/traffic_pmc.txt; rm -rf $O/pf $O/pw
echo "== SQ instruction counters and TA busy of the search kernel (VERDICT r5: none existed for round 5)"
bash profiles/sq_counters_ab.sh mi355x > $O/nn_grid_sq_pmc.txt 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $O/pt -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
python3 profiles/collect_pmc.py $O/pt k_nn_grid > $O/nn_grid_ta_pmc.txt; rm -rf $O/pt
echo "== nn_large: the search kernel at 16M x 16M, kernel stats and traffic PMC"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kn -- python3 profiles/nn_large_only.py > $O/nn_large_16M_under_rocprof.json 2> /dev/null
cp $(find $O/kn -name '*kernel_stats.csv' | head -1) $O/nn_large_16M_kernel_stats.csv; rm -rf $O/kn
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/nf -- python3 profiles/nn_large_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/nw -- python3 profiles/nn_large_only.py > /dev/null 2>&1
python3 profiles/collect_traffic.py $O/nf $O/nw $O/traffic_pmc_nn_large.json > $O/traffic_pmc_nn_large.txt; rm -rf $O/nf $O/nw
echo "== head of a call (hand-written sort)"
rocprofv3 --kernel-trace --output-format csv -d $O/kh -- python3 profiles/call_head_trace.py run > /dev/null 2>&1
python3 profiles/call_head_trace.py analyze $O/kh > $O/call_head_trace.txt; rm -rf $O/kh
echo "== a 28k-point frame: the launches of one estimate(20) in order, on a reused handle and on a fresh one per frame"
python3 profiles/frame_trace.py run > $O/frame_run.txt 2>&1; python3 profiles/frame_trace.py fresh >> $O/frame_run.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/ft -- python3 profiles/frame_trace.py run > /dev/null 2>&1
python3 profiles/frame_trace.py show $O/ft > $O/frame_trace.txt; rm -rf $O/ft
echo "== reference-sized inputs"; python3 bench_small.py > $O/bench_small.txt 2>&1
echo "== virtual ranks: the N-rank orchestration on one GPU (strong: 1M over N ranks; weak: 8 x 1M)"
python3 profiles/multi_virtual_timing.py > $O/multi_virtual_timing.txt 2>&1
python3 profiles/multi_weak_8m.py > $O/multi_weak_8m.txt 2>&1
echo "== two ranks sharing this GPU over gloo + mapped inboxes (functional rehearsal of bench --gpus 2; not a measurement)"
ICP_BENCH_SHARE_GPU=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 2 --brute-steps 1 --weak-steps 20 2> $O/bench_2rank.err | grep '^{' > $O/bench_2rank_shared_ipc.json
echo "== two independent tenants on the one GPU: frame-sized clouds, and converging pairs (one-launch loops in both)"
D=$(mktemp -d); (python3 tests/helpers/independent_handle.py 200 1 $D 2 > $O/tenant1.txt 2>&1 &) ; python3 tests/helpers/independent_handle.py 200 2 $D 2 > $O/tenant2.txt 2>&1; sleep 2
cat $O/tenant1.txt $O/tenant2.txt | grep "^tenant" > $O/two_tenants.txt; rm -f $O/tenant1.txt $O/tenant2.txt
D=$(mktemp -d); (python3 tests/helpers/independent_handle.py 60 1 $D 2 converging > $O/tenant1.txt 2>&1 &) ; python3 tests/helpers/independent_handle.py 60 2 $D 2 converging > $O/tenant2.txt 2>&1; sleep 3
cat $O/tenant1.txt $O/tenant2.txt | grep "^tenant" > $O/two_tenants_converging.txt; rm -f $O/tenant1.txt $O/tenant2.txt
if [ "${1:-}" = "full" ]; then
echo "== sweep: kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kb -- python3 bench.py --nn brute --steps 3 --warmup 1 --cpu-iters 0 --gn-points 0 --nn-points 0 --rotating-calls 0 --converging-calls 0 > $O/bench_brute_1M_under_rocprof.json 2> /dev/null
cp $(find $O/kb -name '*kernel_stats.csv' | head -1) $O/bench_brute_1M_kernel_stats.csv; rm -rf $O/kb
echo "== gn_large: kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kg -- python3 profiles/gn_large_only.py > $O/gn_large_64M_under_rocprof.json 2> /dev/null
cp $(find $O/kg -name '*kernel_stats.csv' | head -1) $O/gn_large_64M_kernel_stats.csv; rm -rf $O/kg
echo "== map"; python3 bench_map.py > $O/bench_map_10M.json 2> /dev/null
fi
ls -la $O
