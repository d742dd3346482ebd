#!/bin/bash
# drop-in call pattern profile + the 12-cell sweep + default bench with CPU legs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
export GSR_PROFILE_OUT=$ROOT/gpurun_out/r03_summary2
mkdir -p $GSR_PROFILE_OUT
cp $ROOT/profiles/traffic.json $GSR_PROFILE_OUT/traffic.json
bash tools/profile_round.sh r03_dropin --unbatched > /dev/null 2>&1
python tools/profile_digest.py r03_dropin > $GSR_PROFILE_OUT/r03_dropin_digest.log 2>&1
cp gpurun_out/r03_dropin/bench_line.json $GSR_PROFILE_OUT/r03_dropin_bench_line.json
rm -rf gpurun_out/r03_dropin/trace gpurun_out/r03_dropin/pmc_*
head -12 $GSR_PROFILE_OUT/r03_dropin_kernel_stats.txt | cut -c1-140
bash tools/sweep.sh r03
timeout 600 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; tail -c 1500 gpurun_out/r03_bench_default.json
