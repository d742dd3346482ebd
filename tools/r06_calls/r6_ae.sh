#!/bin/bash
# call AE: K8 with GsrGrads.zero_outside (clears only the rows the previous writer reached): tests, then the bench A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6ae; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_k8_sparse.py tests/test_graph.py tests/test_views.py tests/test_scratch.py tests/test_epilogue.py tests/test_multirank_gpu.py -x -q -m gpu 2>&1 | tail -15
B="--no-cpu-baseline --no-dropin --sustain-seconds 0 --train-seconds 0"
for r in 1 2; do
  timeout 600 python bench.py $B 2>/dev/null | tail -1 > $O/bench_$r.json
  python - <<PY
import json
d = json.load(open("$O/bench_$r.json"))
print("run $r value", d["value"], "ms", d["ms_per_step"], "rot", d["rotating_cameras"]["by_path_views_per_s"], "stages", d["roofline"].get("stage_us_per_view"))
PY
done
