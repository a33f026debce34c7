#!/bin/bash
# instruction / cycle counts of the LZ4 compress kernel per sequence on one data kind (gpurun, from the repo root):
#   scripts/pmc_runs.sh [dist=runs] [chunks=20000] [dtype=char]
set -e
D=${1:-runs}; N=${2:-20000}; T=${3:-char}
O=gpurun_out/pmc_$D
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rm -rf $O/sq1 $O/sq2
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1 -- python3 scripts/quick_lz4.py --chunks $N --dist $D --dtype $T --reps 1 --count-sequences > $O/sq1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2 -- python3 scripts/quick_lz4.py --chunks $N --dist $D --dtype $T --reps 1 > $O/sq2.log 2>&1
SEQ=$(grep -o "sequences_per_chunk=[0-9.]*" $O/sq1.log | cut -d= -f2)
U=$(python3 -c "print($N*$SEQ)")
echo "# $D $T: $SEQ sequences per chunk; per sequence per wave, quad-cycles"
python3 scripts/pmc_per_window.py $U $O/sq1 $O/sq2
rm -rf $O/sq1 $O/sq2
