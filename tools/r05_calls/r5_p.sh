#!/bin/bash
# call P: the tree as it stands -- full -m gpu suite, smoke(), the default bench line, the rocprofv3 evidence of C3 (kernel trace +
# separate PMC passes, digested on the box), one bench line with the captured graphs forced (rotating cameras through them)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5p; mkdir -p $O; cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -12
timeout 180 python -c "import __graft_entry__ as g; g.smoke()" </dev/null > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
/usr/bin/time -v timeout 600 python bench.py </dev/null > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; grep -E "Elapsed|Maximum resident" $O/bench_default.err
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s", "sustained_views_per_s")}, d["dropin_internal_streams"]["views_per_s"], d["rotating_cameras"]["views_per_s"], d["training_like"]["views_per_s"])
print(d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["max_grad_err_vs_oracle"].get("batched_sum"))
PY
export GSR_PROFILE_OUT=$ROOT/gpurun_out/r05_summary
mkdir -p $GSR_PROFILE_OUT
cp $ROOT/profiles/traffic.json $GSR_PROFILE_OUT/traffic.json 2>/dev/null
bash $ROOT/tools/profile_round.sh r05 --no-dropin > /dev/null 2>&1
(cd $ROOT && python tools/profile_digest.py r05 > $GSR_PROFILE_OUT/r05_digest.log 2>&1)
cp $ROOT/gpurun_out/r05/bench_line.json $GSR_PROFILE_OUT/r05_bench_line.json 2>/dev/null
rm -rf $ROOT/gpurun_out/r05/trace $ROOT/gpurun_out/r05/pmc_*
head -24 $GSR_PROFILE_OUT/r05_kernel_stats.txt | cut -c1-150
cd $ROOT
timeout 300 python bench.py --capture on --no-cpu-baseline --no-dropin --no-roofline --sustain-seconds 0.5 --train-seconds 0.8 </dev/null > $O/bench_capture_on.json 2> $O/bench_capture_on.err
python - <<PY
import json
d = json.loads(open("$O/bench_capture_on.json").read().strip().splitlines()[-1])
print("capture on:", d["value"], d["rotating_cameras"], d["training_like"])
PY
