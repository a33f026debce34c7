#!/bin/bash
# per-window PMC figures of the pair kernel beside the lone-wave mix kernel (knobs build), uniform data as bytes
O=gpurun_out/r5/pmc_pair; mkdir -p $O
N=${1:-20000}
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ3="SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAVES"
export HIPCOMP_PREFETCH=0
for P in 0 1; do
  export HIPCOMP_LZ4_PAIR=$P
  for S in 1 2 3; do
    eval C=\$SQ$S
    rm -rf $O/p${P}_sq$S
    rocprofv3 --kernel-trace --output-format csv --pmc $C -d $O/p${P}_sq$S -- python3 scripts/quick_lz4.py --chunks $N --dist uniform --reps 1 --lib hipcomp-core_amd/lib/libhipcomp_knobs.so > $O/p${P}_sq$S.log 2>&1 || { echo "pass $P $S failed"; tail -5 $O/p${P}_sq$S.log; exit 1; }
  done
  U=$(python3 -c "print($N*1075)")
  { echo "# HIPCOMP_LZ4_PAIR=$P, $N x 64 KiB uniform chunks as bytes, no companion; per 61-byte window (counter / $U); SQ_*CYCLES, SQ_WAIT*, SQ_ACTIVE* in quad-cycles, summed over the waves";
    python3 scripts/pmc_per_window.py $U $O/p${P}_sq1 $O/p${P}_sq2 $O/p${P}_sq3; } > $O/pair${P}.txt
  cat $O/pair${P}.txt
  rm -rf $O/p${P}_sq1 $O/p${P}_sq2 $O/p${P}_sq3
done
