#!/bin/bash
# call Z: kernel trace of the exchange device bench (tools/bench_exchange_device.py): what the row kernels themselves take
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $ROOT/tools/bench_exchange_device.py --gaussians 500000 --res 1024 > $O/run.json 2> $O/run.err; echo "rc=$?"
cd $ROOT
python tools/kstats.py $O/trace > $O/exchange_kernel_stats.txt 2>&1; rm -rf $O/trace
grep -E "k_rows|k_sum_slices|kernel " $O/exchange_kernel_stats.txt | cut -c1-170
