set -e
O=gpurun_out/pmcu
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1 -- python3 scripts/quick_lz4.py --chunks 5000 --dist uniform --reps 1 > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2 -- python3 scripts/quick_lz4.py --chunks 5000 --dist uniform --reps 1 > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc VmemLatency -d $O/l1 -- python3 scripts/quick_lz4.py --chunks 20000 --dist uniform --reps 1 > $O/l1.log 2>&1 || true
python3 scripts/pmc_per_window.py 5375000 $O/sq1 $O/sq2
python3 scripts/pmc_per_window.py 1 $O/l1
