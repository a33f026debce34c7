set -e
O=gpurun_out/pmc_dec
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES"
SQ3="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"
for D in ${DISTS:-harness text}; do
rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_$D -- python3 scripts/quick_lz4.py --chunks 20000 --dist $D --reps 1 --count-sequences > $O/sq1_$D.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2_$D -- python3 scripts/quick_lz4.py --chunks 20000 --dist $D --reps 1 > $O/sq2_$D.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc $SQ3 -d $O/sq3_$D -- python3 scripts/quick_lz4.py --chunks 20000 --dist $D --reps 1 > $O/sq3_$D.log 2>&1 || true
SEQ=$(grep -o "sequences_per_chunk=[0-9.]*" $O/sq1_$D.log | cut -d= -f2)
echo "== $D sequences per chunk $SEQ"
echo "-- decompress"; python3 scripts/pmc_per_window.py --decompress $(python3 -c "print(20000*$SEQ)") $O/sq1_$D $O/sq2_$D $O/sq3_$D
echo "-- compress"; python3 scripts/pmc_per_window.py $(python3 -c "print(20000*$SEQ)") $O/sq1_$D $O/sq2_$D $O/sq3_$D
done
