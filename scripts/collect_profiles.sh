#!/bin/bash
# Collects the profiles of one round on the GPU box (run through gpurun from the repo root):
#   scripts/collect_profiles.sh r02
# kernel stats of the bench command, HBM-side traffic (FETCH / WRITE passes, uniform and harness),
# PMC instruction / cycle counts per window (uniform) and per sequence (harness).
# Output under gpurun_out/prof_$1/; copy what is to be kept into profiles/.
set -e
R=${1:-rXX}
O=gpurun_out/prof_$R
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-variants --no-cpu > $O/bench_under_rocprof.json.log 2>&1
python3 scripts/summarize_profile.py $O/stats $O/${R}_lz4_uniform_char_kernel_stats.csv > /dev/null
for D in uniform harness; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$D -- python3 bench.py --no-cpu --no-variants --steps 2 --warmup 1 --dist $D > $O/fetch_$D.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$D -- python3 bench.py --no-cpu --no-variants --steps 2 --warmup 1 --dist $D > $O/write_$D.log 2>&1
  python3 scripts/hbm_traffic.py $O/fetch_$D $O/write_$D $O/lz4_hbm_traffic.json $D/char/100000 > /dev/null
done
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
# (harness: 20000 chunks, so that the far shape has all its 8192 waves)
for DN in uniform:5000 harness:20000; do
  D=${DN%:*}; N=${DN#*:}
  rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_$D -- python3 scripts/quick_lz4.py --chunks $N --dist $D --reps 1 --count-sequences > $O/sq1_$D.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2_$D -- python3 scripts/quick_lz4.py --chunks $N --dist $D --reps 1 > $O/sq2_$D.log 2>&1
done
{ echo "# lz4 compress kernel (mix shape), 5000 x 64 KiB uniform chunks, per 61-byte window per wave (= per-dispatch counter / 5 375 000 windows); SQ_*CYCLES, SQ_WAIT*, SQ_ACTIVE* are quad-cycles";
  python3 scripts/pmc_per_window.py 5375000 $O/sq1_uniform $O/sq2_uniform; } > $O/${R}_lz4_pmc_per_window_uniform_char.txt
SEQ=$(grep -o "sequences_per_chunk=[0-9.]*" $O/sq1_harness.log | cut -d= -f2)
{ echo "# lz4 compress kernel (far shape), 20000 x 64 KiB harness chunks (300 + (x & 3) int32 as bytes), per LZ4 sequence per wave ($SEQ sequences per chunk); quad-cycles as above";
  python3 scripts/pmc_per_window.py $(python3 -c "print(20000*$SEQ)") $O/sq1_harness $O/sq2_harness; } > $O/${R}_lz4_pmc_per_sequence_harness_char.txt
cat $O/${R}_lz4_pmc_per_window_uniform_char.txt $O/${R}_lz4_pmc_per_sequence_harness_char.txt
tail -c 400 $O/bench_under_rocprof.json.log
