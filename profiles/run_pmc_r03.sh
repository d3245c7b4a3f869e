#!/bin/bash
# round-3: SQ / LDS / TA counters of the search kernels on the benchmark pair (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/a -- python3 profiles/tile_probe.py 5 > /dev/null 2>&1
python3 profiles/collect_pmc.py $O/a k_nn_ > $O/sq1.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $O/b -- python3 profiles/tile_probe.py 5 > /dev/null 2>&1
python3 profiles/collect_pmc.py $O/b k_nn_ > $O/sq2.txt
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $O/c -- python3 profiles/tile_probe.py 5 > /dev/null 2>&1
python3 profiles/collect_pmc.py $O/c k_nn_ > $O/ta.txt
rm -rf $O/a $O/b $O/c
cat $O/sq1.txt $O/sq2.txt $O/ta.txt
