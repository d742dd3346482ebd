#!/bin/bash
# round 4, first call: the new multi-rank tests, then the default bench line of the new bench.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4a; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_multirank_gpu.py -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 400 python bench.py </dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
for k in ("value","ms_per_step","dropin_views_per_s","sustained_views_per_s","training_like","init_state","forward_only"): print(k, d.get(k))
print(d["roofline"]["stage_us_per_view"], d["roofline"]["avg_launch_us"])
PY
