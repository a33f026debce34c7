#!/bin/bash
# Collects the profiles of one round on the GPU box (run through gpurun from the repo root):
#   scripts/collect_profiles.sh r03 [part]      part = headline | rows | pmc | all (default)
#   (ROWS="..." PARTNAME=rowsA: a subset of the rows into $O/${R}_rowsA.json -- a gpurun call is
#   limited to 20 minutes; scripts/merge_rows.py joins the parts)
# headline: rocprofv3 --kernel-trace --stats of the default bench command.
# rows:     per bench row (bench.py extra_keys[].row) kernel trace + stats, and the HBM-side traffic from
#           separate --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles-ready $O/${R}_rows.json
# pmc:      instruction / cycle counts per window (uniform, LDS shape), per sequence (harness, text, runs: far kernels;
#           both decoders on harness and text) and per KiB (Snappy).
# Output under gpurun_out/prof_$R/; copy what is to be kept into profiles/.
set -e
R=${1:-rXX}
PART=${2:-all}
O=gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ $PART = headline ] || [ $PART = all ]; then
  rm -rf $O/stats
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-variants --no-cpu > $O/${R}_bench_under_rocprof.json.log 2> $O/bench_under_rocprof.err
  python3 scripts/summarize_profile.py $O/stats $O/${R}_lz4_uniform_char_kernel_stats.csv > /dev/null
  tail -c 900 $O/${R}_bench_under_rocprof.json.log
fi
if [ $PART = rows ] || [ $PART = all ]; then
  ROWS=${ROWS:-"lz4/uniform/char/100000 lz4/uniform/int/100000 lz4/harness/char/100000 lz4/harness/int/100000 lz4/runs/char/100000 lz4/runs/int/100000 lz4/mixed/char/100000 lz4/misrouted_text_random_samples/char/16384 lz4/misrouted_random_text_samples/char/16384 lz4/misrouted_text_random_first/char/16384 lz4/text/char/65536 lz4/harness/char/1000 lz4/text/char/1000 snappy/text/65536 cascaded/sorted/100000"}
  PARTNAME=${PARTNAME:-rows}
  SPECS=""
  for ROW in $ROWS; do
    K=$(echo $ROW | tr / _)
    rm -rf $O/stats_$K $O/fetch_$K $O/write_$K
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$K -- python3 scripts/run_row.py $ROW --reps 5 > $O/stats_$K.log 2>&1
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$K -- python3 scripts/run_row.py $ROW --reps 2 > $O/fetch_$K.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$K -- python3 scripts/run_row.py $ROW --reps 2 > $O/write_$K.log 2>&1
    python3 scripts/summarize_profile.py $O/stats_$K $O/${R}_kernel_stats_$K.csv > /dev/null
    grep -h "^$ROW" $O/stats_$K.log || true
    SPECS="$SPECS $ROW=$K"
  done
  python3 scripts/profile_rows.py $O $O/${R}_${PARTNAME}.json $SPECS > $O/rows.log 2>&1 || { cat $O/rows.log; exit 1; }
  # (the per-dispatch csv files are large: keep the summaries)
  for ROW in $ROWS; do K=$(echo $ROW | tr / _); rm -rf $O/stats_$K $O/fetch_$K $O/write_$K; done
  tail -n 40 $O/rows.log
fi
if [ $PART = pmc ] || [ $PART = all ]; then
  SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
  SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  # (harness / text: 20000 chunks, so that the far kernel has all its waves)
  for DN in uniform:5000 harness:20000 text:20000 runs:20000; do
    D=${DN%:*}; N=${DN#*:}
    rm -rf $O/sq1_$D $O/sq2_$D
    rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_$D -- python3 scripts/quick_lz4.py --chunks $N --dist $D --reps 1 --count-sequences > $O/sq1_$D.log 2>&1
    rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2_$D -- python3 scripts/quick_lz4.py --chunks $N --dist $D --reps 1 > $O/sq2_$D.log 2>&1
  done
  { echo "# lz4 compress kernel (mix shape), 5000 x 64 KiB uniform chunks, per 61-byte window per wave (= per-dispatch counter / 5 375 000 windows); SQ_*CYCLES, SQ_WAIT*, SQ_ACTIVE* are quad-cycles";
    python3 scripts/pmc_per_window.py 5375000 $O/sq1_uniform $O/sq2_uniform; } > $O/${R}_lz4_pmc_per_window_uniform_char.txt
  for D in harness text runs; do
    SEQ=$(grep -o "sequences_per_chunk=[0-9.]*" $O/sq1_$D.log | cut -d= -f2)
    U=$(python3 -c "print(20000*$SEQ)")
    { echo "# lz4 compress kernel (far kernel), 20000 x 64 KiB $D chunks as bytes, per LZ4 sequence per wave ($SEQ sequences per chunk); quad-cycles as above";
      python3 scripts/pmc_per_window.py $U $O/sq1_$D $O/sq2_$D; } > $O/${R}_lz4_pmc_per_sequence_compress_${D}_char.txt
    { echo "# lz4 decompress kernel, 20000 x 64 KiB $D chunks, per LZ4 sequence per wave ($SEQ sequences per chunk); quad-cycles as above";
      python3 scripts/pmc_per_window.py --decompress $U $O/sq1_$D $O/sq2_$D; } > $O/${R}_lz4_pmc_per_sequence_decompress_${D}.txt
  done
  rm -rf $O/sq1_snappy $O/sq2_snappy
  rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_snappy -- python3 scripts/quick_snappy.py --chunks 16384 --reps 1 > $O/sq1_snappy.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2_snappy -- python3 scripts/quick_snappy.py --chunks 16384 --reps 1 > $O/sq2_snappy.log 2>&1
  U=$((16384*64))
  { echo "# snappy kernels, 16384 x 64 KiB TPC-H-like text, per KiB of input / output per wave; quad-cycles as above";
    echo "-- compress"; python3 scripts/pmc_per_window.py --kernel snappy_compress $U $O/sq1_snappy $O/sq2_snappy;
    echo "-- decompress"; python3 scripts/pmc_per_window.py --kernel snappy_decompress $U $O/sq1_snappy $O/sq2_snappy; } > $O/${R}_snappy_pmc_per_KiB_text.txt
  rm -rf $O/sq1_* $O/sq2_*
  scripts/pmc_cascaded.sh 100000 $O/casc > $O/${R}_cascaded_pmc_per_subchunk.txt 2>&1 || true
  cat $O/${R}_lz4_pmc_per_window_uniform_char.txt $O/${R}_lz4_pmc_per_sequence_*.txt $O/${R}_snappy_pmc_per_KiB_text.txt
fi
