#!/bin/bash
# call AI: zero_outside per arena region (a step with per-view scales does not write the arena's scales region): suite + default bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6ai; mkdir -p $O; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -12
timeout 900 python bench.py </dev/null > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
e = d["max_grad_err_vs_oracle"]
print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s")}, d["roofline"]["frac"])
print("view0", {k: v["max_err_over_max_ref"] for k, v in e["per_tensor"].items()})
print("batched", e["batched_sum"]["arena_sum_max_err_over_max_ref"], {k: v["max_err_over_max_ref"] for k, v in e["batched_sum"]["per_tensor"].items()})
PY
