#!/bin/bash
mkdir -p gpurun_out/r02g
O=gpurun_out/r02g
timeout 600 python -m pytest tests/test_graph.py tests/test_views.py tests/test_context.py tests/test_knn.py tests/test_epilogue.py -m gpu -q 2>&1 | tail -3
for cfg in "c3:" "c3_on:--capture on" "c2:--gaussians 100000 --res 512" "c2_on:--gaussians 100000 --res 512 --capture on"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout 300 python bench.py --no-cpu-baseline $args > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json"))
    c=d["config"]
    print("$name", d["value"], "ms/step", d["ms_per_step"], "dropin", d["dropin_views_per_s"], "enq", d["host_enqueue_ms_per_step"], "wait", d["host_wait_ms_per_step"], c.get("captured_graphs"), c.get("capture_probe"), c.get("capture_stats"))
except Exception as e: print("$name", e)
PY
done
timeout 1500 bash tools/profile_round.sh r02 > $O/profile.log 2>&1; tail -4 $O/profile.log
ls gpurun_out/r02
