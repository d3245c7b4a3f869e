#!/bin/bash
# Everything under profiles/r04_* comes from this script, run on the MI355X box from the repo root:
#   gpurun -- 'bash profiles/collect_r04.sh [full]'
# (counter passes are separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04
mkdir -p $O
B="--brute-steps 0 --cpu-iters 0 --gn-points 0 --converging-calls 0 --rotating-calls 0"
echo "== default bench (what the driver runs)"; python3 bench.py --steps 20 --warmup 5 > $O/bench_default_1M.json 2> $O/bench_default.err
echo "== kernel stats (grid, headline pair)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 5 $B > $O/bench_grid_1M_under_rocprof.json 2> /dev/null
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/bench_grid_1M_kernel_stats.csv; rm -rf $O/kt
echo "== kernel stats (converging pair: the one-launch inner loops)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kc -- python3 profiles/loop_probe.py conv > $O/converging_run.txt 2> /dev/null
cp $(find $O/kc -name '*kernel_stats.csv' | head -1) $O/converging_kernel_stats.csv; rm -rf $O/kc
echo "== traffic PMC"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
python3 profiles/collect_traffic.py $O/pf $O/pw $O/traffic_pmc.json > $O/traffic_pmc.txt; rm -rf $O/pf $O/pw
echo "== TA PMC (search kernels)"
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $O/pt -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
python3 profiles/collect_pmc.py $O/pt k_nn_grid > $O/nn_grid_ta_pmc.txt; rm -rf $O/pt
echo "== SQ instruction counters (search kernel)"; bash profiles/sq_counters_ab.sh mi355x > $O/nn_grid_sq_pmc.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== shared warm walk: in-kernel phase stamps and residency (profiling variant: bash profiles/build_variant.sh coopprof nn_grid.hip -DICP_COOP_PROFILE)"
ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_ab_coopprof.so python3 profiles/coop_phases.py 2>&1 | grep -v amdgpu.ids | tail -24 > $O/search_coop_phases.txt
echo "== virtual ranks: what the N-rank orchestration costs"; python3 profiles/multi_virtual_timing.py > $O/multi_virtual_timing.txt 2>&1
echo "== virtual ranks: kernel trace of one sharded estimate(20)"
for W in 1 8; do
  rocprofv3 --kernel-trace --output-format csv -d $O/km$W -- python3 profiles/multi_trace_count.py run $W > $O/multi_trace_run_$W.txt 2> /dev/null
  python3 profiles/multi_trace_count.py analyze $O/km$W $W > $O/multi_trace_$W.txt; rm -rf $O/km$W
done
cat $O/multi_trace_run_1.txt $O/multi_trace_1.txt $O/multi_trace_run_8.txt $O/multi_trace_8.txt > $O/multi_kernel_trace.txt
echo "== inner-loop launch: in-kernel phase stamps (profiling variant of gn_loop.hip)"
ICP_MI355X_LIB=icp_rust_amd/lib/libicp_ab_loopprof.so python3 profiles/loop_probe.py conv frame 2>&1 | grep -E "^\[loop|converging|frame28k" | awk 'NR % 7 == 1 || /converging|frame28k/' | head -60 > $O/loop_phases.txt
if [ "${1:-}" = "full" ]; then
echo "== sweep: kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kb -- python3 bench.py --nn brute --steps 3 --warmup 1 --cpu-iters 0 --gn-points 0 --rotating-calls 0 --converging-calls 0 > $O/bench_brute_1M_under_rocprof.json 2> /dev/null
cp $(find $O/kb -name '*kernel_stats.csv' | head -1) $O/bench_brute_1M_kernel_stats.csv; rm -rf $O/kb
echo "== gn_large: kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kg -- python3 profiles/gn_large_only.py > $O/gn_large_64M_under_rocprof.json 2> /dev/null
cp $(find $O/kg -name '*kernel_stats.csv' | head -1) $O/gn_large_64M_kernel_stats.csv; rm -rf $O/kg
echo "== reference-sized inputs"; python3 bench_small.py > $O/bench_small.txt 2>&1
echo "== two ranks sharing this GPU over gloo + hipIpc inboxes (functional rehearsal of bench --gpus 2; not a measurement)"
ICP_BENCH_SHARE_GPU=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 2 --brute-steps 0 --weak-steps 0 2> $O/bench_2rank.err | grep '^{' > $O/bench_2rank_shared_ipc.json
echo "== map"; python3 bench_map.py > $O/bench_map_10M.json 2> /dev/null
echo "== search kernel: XCD chunk sizes"; python3 profiles/search_probe.py "" ICP_NN_XCD_CHUNK=0 ICP_NN_XCD_CHUNK=4 ICP_NN_XCD_CHUNK=16 ICP_NN_XCD_CHUNK=64 > $O/search_xcd_chunk.txt 2>&1
fi
ls -la $O
