set -e
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_lz4_gpu.py tests/test_bulk_parity_gpu.py tests/test_golden_gpu.py tests/test_hlif_gpu.py -m gpu -x -q > gpurun_out/r3/gputests3.log 2>&1 || { tail -40 gpurun_out/r3/gputests3.log; exit 1; }
tail -3 gpurun_out/r3/gputests3.log
L=gpurun_out/r3/route2.log
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 20000 --dist uniform,harness,text,runs,mixed auto mix far >> $L 2>&1
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 1000 --dist uniform,harness,text,mixed auto mix far >> $L 2>&1
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 100000 --dist uniform,harness auto >> $L 2>&1
grep -v amdgpu.ids $L
