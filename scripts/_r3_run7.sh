set -e
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_lz4_gpu.py tests/test_bulk_parity_gpu.py tests/test_hlif_gpu.py -m gpu -x -q > gpurun_out/r3/gputests5.log 2>&1 || { tail -40 gpurun_out/r3/gputests5.log; exit 1; }
tail -2 gpurun_out/r3/gputests5.log
timeout -k 10 300 python3 scripts/fuzz_decoders.py > gpurun_out/r3/fuzz1.log 2>&1 || { tail -20 gpurun_out/r3/fuzz1.log; exit 1; }
tail -3 gpurun_out/r3/fuzz1.log
L=gpurun_out/r3/dec1.log
timeout -k 10 600 python3 scripts/quick_lz4.py --chunks 20000 --dist harness,text,runs,uniform --dtype char --reps 3 >> $L 2>&1
timeout -k 10 600 python3 scripts/quick_lz4.py --chunks 100000 --dist harness,runs --dtype char --reps 3 >> $L 2>&1
grep -v amdgpu.ids $L
