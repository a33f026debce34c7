set -e
O=gpurun_out/pmc_text
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- python3 scripts/quick_lz4.py --chunks 20000 --dist text --reps 1 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/l2 -- python3 scripts/quick_lz4.py --chunks 20000 --dist text --reps 1 > $O/l2.log 2>&1 || true
python3 scripts/pmc_per_window.py --kernel "lz4_compress_kernel_far<1, false>" 1 $O/fetch $O/l2
python3 scripts/pmc_per_window.py --decompress 1 $O/fetch $O/l2
