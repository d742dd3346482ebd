#!/bin/bash
mkdir -p gpurun_out/r02f
O=gpurun_out/r02f
NOX=$GRAFT_REPO_ROOT/dreamscene_amd/libgsrast_noxcd.so
timeout 300 python -m pytest tests/test_full_size.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
for cfg in "ranged_xcd::" "ranged_noxcd:GSR_LIB=$NOX:" "lsd4_xcd:GSR_DEPTH_SORT=lsd4:" "lsd4_noxcd:GSR_DEPTH_SORT=lsd4 GSR_LIB=$NOX:" "indoor_cap::--scene indoor --gaussians 2000000 --capture" "indoor_nocap::--scene indoor --gaussians 2000000"; do
  name=${cfg%%:*}; rest=${cfg#*:}; envv=${rest%%:*}; args=${rest#*:}
  case "$args" in *--capture*) a2="${args/--capture/}";; *) a2="$args --no-capture";; esac
  env $envv timeout 300 python bench.py --no-cpu-baseline $a2 > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json"))
    print("$name", d["value"], "ms/step", d["ms_per_step"], "dropin", d["dropin_views_per_s"], d["config"].get("capture_stats"))
    print("   ", d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["stage_us_per_view"])
except Exception as e: print("$name", e)
PY
done
