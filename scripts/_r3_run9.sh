set -e
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_lz4_gpu.py tests/test_bulk_parity_gpu.py -m gpu -x -q > gpurun_out/r3/gputests7.log 2>&1 || { tail -40 gpurun_out/r3/gputests7.log; exit 1; }
tail -2 gpurun_out/r3/gputests7.log
L=gpurun_out/r3/easy1.log
timeout -k 10 600 python3 scripts/quick_lz4.py --chunks 20000 --dist harness,text --dtype char,int --reps 3 >> $L 2>&1
timeout -k 10 600 python3 scripts/quick_lz4.py --chunks 100000 --dist harness --dtype char,int --reps 3 >> $L 2>&1
grep -v amdgpu.ids $L
