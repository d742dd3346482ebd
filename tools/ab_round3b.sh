#!/bin/bash
# one gpurun call: parity suite on the new tree, then interleaved A/B legs (scratch contract; sort tile size)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/$1; mkdir -p $O; cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -12
B="--no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0"
for r in 1 2 3; do
  for v in clean legacy os8; do
    unset GSR_LIB
    if [ $v = os8 ]; then export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_os8.so; m=clean; else m=$v; fi
    timeout 300 python tools/ab_scratch.py $m $B </dev/null > $O/bench_${v}_$r.json 2>$O/bench_${v}_$r.err
    python - <<PY
import json
try:
    d=json.load(open("$O/bench_${v}_$r.json")); print("$v $r", d["value"], d["roofline"]["stage_us_per_view"])
except Exception as e: print("$v $r failed", e)
PY
  done
done
unset GSR_LIB
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print({k:d[k] for k in ('value','ms_per_step','dropin_views_per_s','sustained_views_per_s')}, d['rotating_cameras'] and d['rotating_cameras']['views_per_s'])"
