#!/bin/bash
# Memory-side traffic of the Snappy kernels (run through gpurun from the repo root): FETCH_SIZE / WRITE_SIZE
# (KiB per dispatch; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes) and the L2's hit / miss counts.
set -e
O=gpurun_out/pmc_snappy_traffic
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- python3 scripts/quick_snappy.py --chunks 16384 --reps 1 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -- python3 scripts/quick_snappy.py --chunks 16384 --reps 1 > $O/write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/l2 -- python3 scripts/quick_snappy.py --chunks 16384 --reps 1 > $O/l2.log 2>&1 || true
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $O/l1 -- python3 scripts/quick_snappy.py --chunks 16384 --reps 1 > $O/l1.log 2>&1 || true
for k in snappy_compress snappy_decompress; do
  python3 scripts/pmc_per_window.py --kernel $k 1 $O/fetch $O/write $O/l2 $O/l1
done
