set -e
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_lz4_gpu.py -m gpu -x -q -k "compressible or repetitive or edge" > gpurun_out/r3/gputests4.log 2>&1 || { tail -40 gpurun_out/r3/gputests4.log; exit 1; }
tail -2 gpurun_out/r3/gputests4.log
L=gpurun_out/r3/route3.log
timeout -k 10 300 python3 scripts/route_counts.py >> $L 2>&1
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 20000 --dist uniform,mixed auto mix >> $L 2>&1
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 1000 --dist uniform auto mix >> $L 2>&1
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 100000 --dist uniform,harness auto mix >> $L 2>&1
grep -v amdgpu.ids $L
bash scripts/pmc_sequences.sh > gpurun_out/r3/pmc_dec.log 2>&1
cat gpurun_out/r3/pmc_dec.log | grep -v amdgpu.ids
