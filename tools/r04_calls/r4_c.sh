#!/bin/bash
# round 4, call c: fp64 cross-wave sums -- parity suite, run-to-run spread, cost (A/B against the round-3 library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4c; mkdir -p $O; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -15
timeout 600 python tools/determinism_probe.py 8 > $O/determinism.json 2> $O/determinism.err; echo "probe rc=$?"; cat $O/determinism.json | head -150
for r in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --rotate-seconds 0 --train-seconds 0 --sustain-seconds 1 --no-dropin </dev/null > $O/b.json 2>$O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json")); print("new $r", d["value"], d["roofline"]["stage_us_per_view"], d["roofline"]["avg_launch_us"])
except Exception as e: print("new $r failed", e)
PY
done
