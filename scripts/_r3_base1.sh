set -e
mkdir -p gpurun_out/r3
for N in 500 1000 4000; do
 for SH in mix far; do
  echo "== chunks $N shape $SH" >> gpurun_out/r3/base1.log
  HIPCOMP_LZ4_SHAPE=$SH timeout -k 10 300 python3 scripts/quick_lz4.py --chunks $N --dist harness,text --dtype char --reps 3 >> gpurun_out/r3/base1.log 2>&1
 done
done
echo "== chunks 1000 farw runs" >> gpurun_out/r3/base1.log
HIPCOMP_LZ4_SHAPE=farw timeout -k 10 300 python3 scripts/quick_lz4.py --chunks 1000 --dist runs --dtype char,int --reps 3 >> gpurun_out/r3/base1.log 2>&1
echo "== chunks 20000 auto" >> gpurun_out/r3/base1.log
timeout -k 10 300 python3 scripts/quick_lz4.py --chunks 20000 --dist harness,text,runs --dtype char --reps 3 >> gpurun_out/r3/base1.log 2>&1
cat gpurun_out/r3/base1.log
