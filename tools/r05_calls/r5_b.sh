#!/bin/bash
# call B: early pair count (kOsEarlyN) + internal streams: the suite, the per-view interface A/B, the headline
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5b; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -15
timeout 300 python tools/bench_dropin.py --gaussians 500000 --res 1024 --seconds 0.8 --patterns fb4,one_bw,fwd > $O/dropin_c3.txt 2>&1; echo "dropin c3 rc=$?"; grep -E '^\{|^SIDE' $O/dropin_c3.txt | cut -c1-200
timeout 300 python tools/bench_dropin.py --gaussians 100000 --res 512 --seconds 0.8 --patterns fb4,one_bw,fwd > $O/dropin_c2.txt 2>&1; echo "dropin c2 rc=$?"; grep -E '^\{|^SIDE' $O/dropin_c2.txt | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --train-seconds 0 --rotate-seconds 0 </dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s", "sustained_views_per_s", "host_wait_ms_per_step", "host_busy_ms_per_step")})
    r = d["roofline"]; print(r["stage_us_per_view"])
except Exception as e:
    print("no bench line:", e)
PY
