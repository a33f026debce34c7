set -e
mkdir -p gpurun_out/r3
L=gpurun_out/r3/both2.log
timeout -k 10 300 python3 scripts/trip_stats.py --chunks 2000 --dist text,harness --config far >> $L 2>&1
timeout -k 10 300 python3 scripts/trip_stats.py --chunks 1000 --dist text --config far:4,0,2048 >> $L 2>&1
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 20000 --dist text auto auto:1,3,512 auto:1,2,512 auto:1,1,512 auto:1,4,512 auto:1,5,512 auto:2,4,512 auto:1,3,256 auto:1,3,1024 >> $L 2>&1
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 100000 --dist harness auto auto:1,7,512 >> $L 2>&1
grep -v amdgpu.ids $L
