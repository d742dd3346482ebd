#!/bin/bash
# call G: the final tree -- suite, smoke, the C3 profile set (trace + PMC), the 12-cell sweep, the default bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5g; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -12
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" </dev/null > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
export GSR_PROFILE_OUT=$ROOT/gpurun_out/r05_summary
mkdir -p $GSR_PROFILE_OUT
cp $ROOT/profiles/traffic.json $GSR_PROFILE_OUT/traffic.json 2>/dev/null
bash $ROOT/tools/profile_round.sh r05 --no-dropin > /dev/null 2>&1
python tools/profile_digest.py r05 > $GSR_PROFILE_OUT/r05_digest.log 2>&1
cp $ROOT/gpurun_out/r05/bench_line.json $GSR_PROFILE_OUT/r05_bench_line.json 2>/dev/null
rm -rf $ROOT/gpurun_out/r05/trace $ROOT/gpurun_out/r05/pmc_*
head -20 $GSR_PROFILE_OUT/r05_kernel_stats.txt | cut -c1-150
bash tools/sweep.sh r05
timeout 500 python bench.py </dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s", "sustained_views_per_s", "host_wait_ms_per_step", "host_busy_ms_per_step", "init_views_per_s")})
    print("streams:", d["dropin_internal_streams"]["views_per_s"]); print("fwd:", d["forward_only"]["dropin_views_per_s"], d["forward_only"]["batched_views_per_s"], d["forward_only"]["score_views"]); print("train:", d["training_like"]["views_per_s"], d["training_like"]["dropin_views_per_s"]); print("rot:", d["rotating_cameras"]["views_per_s"])
    r = d["roofline"]; print({k: r[k] for k in ("kernel", "frac", "traffic", "traffic_kernel", "avg_launch_us")}); print(r["stage_us_per_view"]); print(r["whole_path"]["launch_accurate_frac_of_hbm_peak"], r["whole_path"]["counter_frac_of_hbm_peak"])
    print(d["config"]["batched_through"], d["config"]["capture_probe"])
    print(json.dumps(d["max_grad_err_vs_oracle"]["batched_sum"])[:400])
except Exception as e:
    print("no bench line:", e)
PY
