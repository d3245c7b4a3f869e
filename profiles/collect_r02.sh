#!/bin/bash
# Everything under profiles/r02_* comes from this script, run on the MI355X box from the repo root:
#   gpurun -- 'bash profiles/collect_r02.sh'
# (counter passes are separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
B="--brute-steps 0 --cpu-iters 0 --gn-points 0"
echo "== default bench"; python3 bench.py > $O/bench_default_1M.json 2> $O/bench_default.err
echo "== kernel stats (grid)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py $B > $O/bench_grid_1M_under_rocprof.json 2> /dev/null
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/bench_grid_1M_kernel_stats.csv; rm -rf $O/kt
echo "== traffic PMC"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
python3 profiles/collect_traffic.py $O/pf $O/pw $O/traffic_pmc.json > $O/traffic_pmc.txt; rm -rf $O/pf $O/pw
echo "== SQ PMC (search kernels)"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/ps -- python3 bench.py --steps 40 --warmup 2 $B > /dev/null 2>&1
python3 profiles/collect_pmc.py $O/ps k_nn_grid > $O/nn_grid_sq_pmc.txt; rm -rf $O/ps
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $O/pt -- python3 bench.py --steps 40 --warmup 2 $B > $O/ta_pmc.log 2>&1
python3 profiles/collect_pmc.py $O/pt k_nn_grid > $O/nn_grid_ta_pmc.txt; rm -rf $O/pt
echo "== sweep: kernel stats + PMC"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kb -- python3 bench.py --nn brute --steps 3 --warmup 1 --cpu-iters 0 --gn-points 0 > $O/bench_brute_1M_under_rocprof.json 2> /dev/null
cp $(find $O/kb -name '*kernel_stats.csv' | head -1) $O/bench_brute_1M_kernel_stats.csv; rm -rf $O/kb
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/pb -- python3 bench.py --nn brute --steps 2 --warmup 1 --cpu-iters 0 --gn-points 0 > /dev/null 2>&1
python3 profiles/collect_pmc.py $O/pb k_nn_brute > $O/brute_pmc.txt; rm -rf $O/pb
ICP_NN_OLD_SCREEN=1 python3 bench.py --nn brute --steps 3 --warmup 1 --cpu-iters 0 --gn-points 0 > $O/bench_brute_1M_old_screen.json 2>/dev/null
echo "== reference-sized inputs"; python3 bench_small.py > $O/bench_small.txt 2>&1
echo "== pipelined scan3d frame loop: kernel timeline"
rocprofv3 --kernel-trace --output-format csv -d $O/kp -- python3 profiles/scan3d_timeline.py > $O/scan3d_timeline.log 2>&1
python3 profiles/scan3d_timeline.py --analyze $O/kp > $O/timeline_scan3d_pipelined.txt 2>&1; rm -rf $O/kp
echo "== two ranks sharing this GPU over gloo (functional rehearsal of bench --gpus 2; not a measurement)"
ICP_BENCH_SHARE_GPU=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 40 --warmup 2 --brute-steps 0 --weak-steps 20 2> /dev/null | grep '^{' > $O/bench_2rank_shared_gloo.json
echo "== virtual ranks: cost of the exchanges"; python3 profiles/multi_virtual_timing.py > $O/multi_virtual_timing.txt 2>&1
echo "== map"; python3 bench_map.py > $O/bench_map_10M.json 2> /dev/null
ls -la $O
