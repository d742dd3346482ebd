#!/bin/bash
# same-call A/B of two builds of libgsrast: interleaved bench runs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/$1; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_fuzz.py tests/test_gpu_parity.py tests/test_full_size.py tests/test_score_views.py -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for r in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export GSR_LIB=$ROOT/dreamscene_amd/$2; else unset GSR_LIB; fi
    timeout 300 python bench.py --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 </dev/null > $O/bench_${v}_$r.json 2>$O/bench_${v}_$r.err
    python - <<PY
import json
d=json.load(open("$O/bench_${v}_$r.json")); print("$v $r", d["value"], d["roofline"]["stage_us_per_view"])
PY
  done
done
