#!/bin/bash
# round 3: the evidence behind DESIGN.md section 5 "k_nn_tile" (the LDS-tile warm search, ICP_NN_TILE=1) -- per-search
# time, in-kernel phase stamps (diagnostic build: make -C icp_rust_amd/csrc stats), kernel stats, SQ / LDS / TA counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tile; rm -rf $O; mkdir -p $O
export ICP_NN_TILE=1
python3 profiles/tile_probe.py 2>&1 | grep -v amdgpu > $O/probe.txt
ICP_MI355X_LIB=icp_rust_amd/lib/libicp_mi355x_stats.so python3 profiles/tile_phases.py 2>&1 | grep -v amdgpu > $O/phases.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 profiles/tile_probe.py > /dev/null 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/kt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/a -- python3 profiles/tile_probe.py 5 > /dev/null 2>&1
python3 profiles/collect_pmc.py $O/a k_nn_ > $O/sq.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $O/b -- python3 profiles/tile_probe.py 5 > /dev/null 2>&1
python3 profiles/collect_pmc.py $O/b k_nn_ > $O/sq_lds.txt
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $O/c -- python3 profiles/tile_probe.py 5 > /dev/null 2>&1
python3 profiles/collect_pmc.py $O/c k_nn_ > $O/ta.txt
rm -rf $O/a $O/b $O/c
cat $O/probe.txt $O/phases.txt
