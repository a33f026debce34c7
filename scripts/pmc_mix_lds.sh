#!/bin/bash
set -e
O=gpurun_out/r4/pmc_mix
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P1="GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
i=0
for P in "$P1" "$P2"; do i=$((i+1)); rm -rf $O/p$i
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $P -d $O/p$i -- python3 scripts/quick_lz4.py --chunks 5000 --dist uniform --reps 1 > $O/p$i.log 2>&1 || { tail -3 $O/p$i.log; echo fail $i; }
done
echo "# per 61-byte window per wave (5 375 000 windows)"
python3 scripts/pmc_per_window.py 5375000 $O/p1 $O/p2
rm -rf $O/p1 $O/p2
