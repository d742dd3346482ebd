#!/bin/bash
# the default bench line, the C3 profile set, the sweep and a drop-in trace of the final tree
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 200 python bench.py 2> gpurun_out/r04_bench_default.err | tail -1 > gpurun_out/r04_bench_default.json
python -c "import json; d=json.load(open('gpurun_out/r04_bench_default.json')); print({k: d.get(k) for k in ('value','ms_per_step','dropin_views_per_s')}, d['config'].get('seg_len'), d['roofline']['frac'])"
timeout 150 bash tools/profile_all.sh r04 c3 2>&1 | tail -16
timeout 150 bash tools/sweep.sh r04 2>&1 | tail -12
export GSR_PROFILE_OUT=$ROOT/gpurun_out/r04_summary
timeout 100 bash tools/profile_round.sh r04_dropin --unbatched --no-dropin > /dev/null 2>&1
python tools/profile_digest.py r04_dropin > $GSR_PROFILE_OUT/r04_dropin_digest.log 2>&1
cp gpurun_out/r04_dropin/bench_line.json $GSR_PROFILE_OUT/r04_dropin_bench_line.json 2>/dev/null
rm -rf gpurun_out/r04_dropin/trace gpurun_out/r04_dropin/pmc_* gpurun_out/r04/trace gpurun_out/r04/pmc_*
head -12 $GSR_PROFILE_OUT/r04_dropin_kernel_stats.txt | cut -c1-150
du -sh gpurun_out | tail -1
