#!/bin/bash
# call AJ: why is the captured path slower than the eager one with rotating cameras? its capture statistics
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6aj; mkdir -p $O; cd $ROOT
timeout 600 python bench.py --no-cpu-baseline --no-dropin --sustain-seconds 0 --train-seconds 0 --no-roofline --rotate-seconds 4 2>/dev/null | tail -1 > $O/bench.json
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("value", d["value"], json.dumps(d["rotating_cameras"]))
PY
