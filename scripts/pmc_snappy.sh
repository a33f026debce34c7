#!/bin/bash
# PMC instruction / cycle counts of the Snappy kernels per KiB of input (run through gpurun from the repo root).
set -e
O=gpurun_out/pmc_snappy
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES"
rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1 -- python3 scripts/quick_snappy.py --chunks 16384 --reps 1 > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2 -- python3 scripts/quick_snappy.py --chunks 16384 --reps 1 > $O/sq2.log 2>&1
U=$((16384*64))
echo "-- snappy compress, per KiB of input per wave"; python3 scripts/pmc_per_window.py --kernel snappy_compress $U $O/sq1 $O/sq2
echo "-- snappy decompress, per KiB of output per wave"; python3 scripts/pmc_per_window.py --kernel snappy_decompress $U $O/sq1 $O/sq2
