#!/bin/bash
mkdir -p gpurun_out/r02j
O=gpurun_out/r02j
AB=$GRAFT_REPO_ROOT/dreamscene_amd/libgsrast_ab.so
timeout 600 python -m pytest tests/test_views.py tests/test_graph.py tests/test_scene.py tests/test_epilogue.py tests/test_context.py tests/test_golden.py -m gpu -q 2>&1 | tail -3
for cfg in "new::" "old:GSR_LIB=$AB:" "new_init::--init-opacity" "new_indoor::--scene indoor --gaussians 2000000" "old_indoor:GSR_LIB=$AB:--scene indoor --gaussians 2000000" "new_2m::--gaussians 2000000 --res 512" "old_2m:GSR_LIB=$AB:--gaussians 2000000 --res 512"; do
  name=${cfg%%:*}; rest=${cfg#*:}; envv=${rest%%:*}; args=${rest#*:}
  env $envv timeout 300 python bench.py --no-cpu-baseline --no-dropin --capture off --steps 100 $args > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json"))
    print("$name", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["stage_us_per_view"])
except Exception as e: print("$name", e)
PY
done
