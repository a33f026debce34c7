#!/bin/bash
# per-sequence PMC figures of the LZ4 decoder of two builds on one kind of data: pmc_dec_ab.sh DIST LIB_A LIB_B
O=gpurun_out/r5/pmc_dec; mkdir -p $O
D=${1:-text}; shift
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_SMEM"
for L in "$@"; do
  for S in 1 2; do
    eval C=\$SQ$S
    rm -rf $O/${L}_sq$S
    rocprofv3 --kernel-trace --output-format csv --pmc $C -d $O/${L}_sq$S -- python3 scripts/quick_lz4.py --chunks 20000 --dist $D --reps 1 --lib hipcomp-core_amd/lib/libhipcomp_$L.so > $O/${L}_sq$S.log 2>&1 || { echo "pass $L $S failed"; tail -5 $O/${L}_sq$S.log; exit 1; }
  done
  { echo "# $L, 20000 x 64 KiB $D chunks: decoder, per KiB of output (counter / 1 310 720)"; python3 scripts/pmc_per_window.py --decompress 1310720 $O/${L}_sq1 $O/${L}_sq2; } > $O/dec_${D}_$L.txt
  cat $O/dec_${D}_$L.txt
  rm -rf $O/${L}_sq1 $O/${L}_sq2
done
