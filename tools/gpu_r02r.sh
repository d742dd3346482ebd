#!/bin/bash
# two-launch sparse K8 (classify + listed), A/B against the dense kernels (GSR_K8_SPARSE=0)
mkdir -p gpurun_out/r02r
O=gpurun_out/r02r
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_views.py tests/test_graph.py tests/test_fuzz.py tests/test_epilogue.py tests/test_context.py tests/test_scene.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -8
for cfg in "sparse::" \
           "sparse_init::--init-opacity --no-dropin" \
           "sparse_indoor::--scene indoor --gaussians 2000000 --no-dropin" \
           "sparse_2m::--gaussians 2000000 --res 512 --no-dropin" \
           "sparse_c2::--gaussians 100000 --res 512 --capture on" "sparse_250k::--gaussians 250000 --res 800" ; do
  name=${cfg%%:*}; rest=${cfg#*:}; envv=${rest%%:*}; args=${rest#*:}
  env $envv timeout 300 python bench.py </dev/null --no-cpu-baseline --capture off --steps 100 $args > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json"))
    print("$name", d["value"], "dropin", d.get("dropin_views_per_s"), "ms/step", d["ms_per_step"], d["roofline"]["stage_us_per_view"])
except Exception as e: print("$name", e)
PY
done
