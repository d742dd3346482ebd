#!/bin/bash
# call AG: zero_outside with the trusted / untrusted graph pair and the union bitmap behind the message exchanges: whole suite + bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6ag; mkdir -p $O; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -12
timeout 600 python tools/fuzz_rowmsg.py 100 0 2>&1 | tail -2
B="--no-cpu-baseline --sustain-seconds 0"
for r in 1 2; do
  timeout 600 python bench.py $B 2>/dev/null | tail -1 > $O/bench_$r.json
  python - <<PY
import json
d = json.load(open("$O/bench_$r.json"))
print("run $r value", d["value"], "ms", d["ms_per_step"], "dropin", d["dropin_views_per_s"], "rot", d["rotating_cameras"]["by_path_views_per_s"], "tl", d["training_like"]["by_path_views_per_s"], "trainer", json.dumps(d.get("trainer_step"))[:300], d["config"].get("capture_stats"))
PY
done
