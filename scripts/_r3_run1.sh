set -e
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3/gputests1.log 2>&1 || { tail -40 gpurun_out/r3/gputests1.log; exit 1; }
tail -5 gpurun_out/r3/gputests1.log
L=gpurun_out/r3/both1.log
timeout -k 10 300 python3 scripts/ab_shapes.py --chunks 1000 --dist harness,text far far:4,0,2048 far:1,0,2048 >> $L 2>&1
timeout -k 10 300 python3 scripts/ab_shapes.py --chunks 1000 --dist runs --dtype char,int farw farw:4,0,2048 >> $L 2>&1
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 20000 --dist harness,text auto auto:0,4,2048 auto:0,4,512 auto:4,0,2048 auto:4,12,1024 auto:1,3,512 auto:1,7,512 auto:1,7,1024 auto:2,6,1024 >> $L 2>&1
timeout -k 10 600 python3 scripts/ab_shapes.py --chunks 20000 --dist runs --dtype char,int auto auto:4,12,1024 auto:1,7,512 >> $L 2>&1
grep -v amdgpu.ids $L
