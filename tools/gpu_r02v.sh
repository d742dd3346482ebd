#!/bin/bash
# host-side numbers of the captured step at C3, and the training-step benchmark on the final kernels
O=gpurun_out/r02v; mkdir -p $O
timeout 120 python bench.py </dev/null --no-cpu-baseline --no-dropin --capture on --steps 100 > $O/c3_captured.json 2> $O/c3_captured.err
python - <<PY
import json
d=json.load(open("$O/c3_captured.json"))
print({k:d[k] for k in ("value","ms_per_step","host_enqueue_ms_per_step","host_wait_ms_per_step","host_busy_ms_per_step")}, d["config"]["capture_stats"])
PY
timeout 150 python tools/bench_train_step.py </dev/null > $O/train_step.log 2>&1; tail -8 $O/train_step.log
