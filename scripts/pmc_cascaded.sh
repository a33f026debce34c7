#!/bin/bash
# PMC instruction / cycle counts of the Cascaded kernels per 4096-byte sub-chunk per wave (run through gpurun from the repo root).
set -e
O=gpurun_out/pmc_cascaded
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1 -- python3 scripts/quick_cascaded.py --parts 20000 --reps 1 > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2 -- python3 scripts/quick_cascaded.py --parts 20000 --reps 1 > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc LdsLatency -d $O/l1 -- python3 scripts/quick_cascaded.py --parts 20000 --reps 1 > $O/l1.log 2>&1 || true
rocprofv3 --kernel-trace --output-format csv --pmc VmemLatency -d $O/l2 -- python3 scripts/quick_cascaded.py --parts 20000 --reps 1 > $O/l2.log 2>&1 || true
U=$((20000*16))
echo "-- cascaded compress, per 4 KiB sub-chunk"; python3 scripts/pmc_per_window.py --kernel cascaded_compress $U $O/sq1 $O/sq2
echo "-- cascaded decompress, per 4 KiB sub-chunk"; python3 scripts/pmc_per_window.py --kernel cascaded_decompress $U $O/sq1 $O/sq2
python3 scripts/pmc_per_window.py --kernel cascaded_compress 1 $O/l1 $O/l2
python3 scripts/pmc_per_window.py --kernel cascaded_decompress 1 $O/l1 $O/l2
