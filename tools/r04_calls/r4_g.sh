#!/bin/bash
# round 4, call g: K1 in two phases on two streams: parity, then A/B against GSR_K1_SPLIT=0 (batched + drop-in, C3 and 100k)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4g; mkdir -p $O; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -15
for cfg in "" "--gaussians 100000 --res 512" "--init-opacity"; do
for r in 1 2; do
for v in 1 0; do
  GSR_K1_SPLIT=$v timeout 300 python bench.py --no-cpu-baseline --rotate-seconds 0 --train-seconds 0 --sustain-seconds 1 $cfg </dev/null > $O/b.json 2>$O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json")); print("split=$v $r $cfg", d["value"], d["sustained_views_per_s"], "dropin", d["dropin_views_per_s"], d["config"]["batched_through"][:20], d["roofline"]["stage_us_per_view"])
except Exception as e: print("failed", e); print(open("$O/b.err").read()[-1500:])
PY
done
done
done
