#!/bin/bash
# K8 sparse for K >= 9 only: whole suite again; coalesced dL/dmeans2D clears A/B; then the sweep of SURVEY.md 8(d)
O=gpurun_out/r02u; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
M2=$GRAFT_REPO_ROOT/dreamscene_amd/libgsrast_m2.so
for cfg in "base::" "m2:GSR_LIB=$M2:" "base_2m::--gaussians 2000000 --res 512" "m2_2m:GSR_LIB=$M2:--gaussians 2000000 --res 512" "indoor::--scene indoor --gaussians 2000000"; do
  name=${cfg%%:*}; rest=${cfg#*:}; envv=${rest%%:*}; args=${rest#*:}
  env $envv timeout 120 python bench.py </dev/null --no-cpu-baseline --no-dropin --capture off --steps 100 $args > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json"))
    print("$name", d["value"], d["roofline"]["stage_us_per_view"]["preprocess_bwd"])
except Exception as e: print("$name", e)
PY
done
timeout 260 bash tools/sweep.sh r02 </dev/null
