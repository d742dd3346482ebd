#!/bin/bash
# call Y: the suite, smoke() and the default bench line on the tree after calls Q..X
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5y; mkdir -p $O; cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -12
timeout 180 python -c "import __graft_entry__ as g; g.smoke()" </dev/null > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py </dev/null > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s", "sustained_views_per_s")}, d["dropin_internal_streams"]["views_per_s"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
PY
