set -e
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_snappy_gpu.py tests/test_bulk_parity_gpu.py tests/test_hlif_gpu.py tests/test_golden_gpu.py -m gpu -x -q > gpurun_out/r3/gputests6.log 2>&1 || { tail -40 gpurun_out/r3/gputests6.log; exit 1; }
tail -2 gpurun_out/r3/gputests6.log
timeout -k 10 300 python3 scripts/fuzz_decoders.py > gpurun_out/r3/fuzz2.log 2>&1 || { tail -20 gpurun_out/r3/fuzz2.log; exit 1; }
tail -3 gpurun_out/r3/fuzz2.log
timeout -k 10 600 python3 scripts/quick_snappy.py --chunks 16384 --reps 3 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python3 scripts/quick_snappy.py --chunks 65536 --reps 3 2>&1 | grep -v amdgpu.ids
