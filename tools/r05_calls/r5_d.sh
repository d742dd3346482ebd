#!/bin/bash
# call D: the suite on the tree with the LDS-counting score kernel, the score paths, the default bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5d; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -15
timeout 300 python tools/bench_score.py 500000 1024 > $O/score_c3.json 2> $O/score.err; echo "score rc=$?"; cut -c1-700 $O/score_c3.json
timeout 300 python tools/bench_score.py 100000 512 > $O/score_c2.json 2>> $O/score.err; cut -c1-700 $O/score_c2.json
timeout 500 python bench.py </dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s", "sustained_views_per_s", "host_wait_ms_per_step", "host_busy_ms_per_step", "init_views_per_s")})
    print("streams:", d["dropin_internal_streams"]); print("fwd:", d["forward_only"]); print("train:", d["training_like"]); print("rot:", d["rotating_cameras"]["views_per_s"])
    r = d["roofline"]; print({k: r[k] for k in ("kernel", "frac", "traffic", "traffic_kernel", "avg_launch_us")}); print(r["stage_us_per_view"])
    print(json.dumps(d["max_grad_err_vs_oracle"]["batched_sum"])[:600])
except Exception as e:
    print("no bench line:", e)
PY
