#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4d; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_dropin_graphs.py tests/test_graph.py tests/test_context.py -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.log
for cfg in "" "--gaussians 100000 --res 512"; do
for v in 1 0; do
  GSR_DROPIN_GRAPHS=$v timeout 300 python bench.py --no-cpu-baseline --rotate-seconds 0 --sustain-seconds 1 $cfg </dev/null > $O/b.json 2>$O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json")); print("graphs=$v $cfg", d["value"], "dropin", d["dropin_views_per_s"], "train", d["training_like"], "fwd", d["forward_only"], d["dropin_graphs"])
except Exception as e: print("failed", e); print(open("$O/b.err").read()[-2000:])
PY
done
done
