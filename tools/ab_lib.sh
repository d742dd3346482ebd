#!/bin/bash
# one gpurun call: quick parity on the new tree, then interleaved A/B bench legs of the product library against a variant
# usage: tools/ab_lib.sh <tag> <variant .so under dreamscene_amd/> [extra bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/$1; mkdir -p $O; cd $ROOT
VAR=$2; shift 2
timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -12
for cfg in "" "--init-opacity" "--gaussians 2000000 --res 512"; do
  echo "== config: $cfg $@"
  for r in 1 2; do
    for v in new old; do
      if [ $v = old ]; then export GSR_LIB=$ROOT/dreamscene_amd/$VAR; else unset GSR_LIB; fi
      timeout 300 python bench.py --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 $cfg "$@" </dev/null > $O/b.json 2>$O/b.err
      python - <<PY
import json
try:
    d=json.load(open("$O/b.json")); print("$v $r", d["value"], d["roofline"]["stage_us_per_view"])
except Exception as e: print("$v $r failed", e)
PY
    done
  done
done
