set -e
mkdir -p gpurun_out/r3
L=gpurun_out/r3/both3.log
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 20000 --dist text auto:1,3,512 auto:1,3,512@n52 auto:1,3,512@n52f32 auto:1,3,512@n52f24 auto:1,2,512@n52 >> $L 2>&1
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 1000 --dist text,harness far:1,0,2048 far:1,0,2048@n52 >> $L 2>&1
grep -v amdgpu.ids $L
