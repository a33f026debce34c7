set -e
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_snappy_gpu.py tests/test_bulk_parity_gpu.py tests/test_golden_gpu.py -m gpu -x -q -k "snappy or Snappy" > gpurun_out/r3/gputests8.log 2>&1 || { tail -40 gpurun_out/r3/gputests8.log; exit 1; }
tail -2 gpurun_out/r3/gputests8.log
timeout -k 10 600 python3 scripts/quick_snappy.py --chunks 65536 --reps 3 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python3 scripts/parity_sweep_snappy.py 2>&1 | tail -3
