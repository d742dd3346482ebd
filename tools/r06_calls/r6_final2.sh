#!/bin/bash
# final call (2): suite + smoke + default bench line of the final tree
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6final2; mkdir -p $O; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -8
timeout 180 python -c "import __graft_entry__ as g; g.smoke()" </dev/null > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
( time timeout 900 python bench.py </dev/null > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; tail -3 $O/bench_time.txt; tail -2 $O/bench_default.err
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
e = d["max_grad_err_vs_oracle"]
print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s")}, d["roofline"]["frac"], d["cpu_baseline"]["value"], "parity", e["within_1e-5_of_own_scale"])
print("rot", d["rotating_cameras"]["by_path_views_per_s"], "tl", d["training_like"]["by_path_views_per_s"])
PY
