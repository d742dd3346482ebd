#!/bin/bash
mkdir -p gpurun_out/r02e
O=gpurun_out/r02e
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
for cfg in "nocap:--no-capture" "cap:" "nocap_lsd4:--no-capture" "cap_lsd4:" "c2:--gaussians 100000 --res 512" "c2_nocap:--gaussians 100000 --res 512 --no-capture" "init:--init-opacity" "v1:--views-per-step 1" "indoor:--scene indoor --gaussians 2000000"; do
  name=${cfg%%:*}; args=${cfg#*:}
  envv=""; case $name in *lsd4) envv="GSR_DEPTH_SORT=lsd4";; esac
  env $envv timeout 300 python bench.py --no-cpu-baseline $args > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json"))
    print("$name", d["value"], "ms/step", d["ms_per_step"], "dropin", d["dropin_views_per_s"], "enq", d["host_enqueue_ms_per_step"], "wait", d["host_wait_ms_per_step"], d["config"].get("capture_stats"), (d.get("exchange") or {}).get("nonzero_row_frac"))
    print("   ", d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["stage_us_per_view"])
except Exception as e: print("$name", e)
PY
done
timeout 200 python tools/host_profile_dropin.py > $O/host_dropin.txt 2>&1; head -45 $O/host_dropin.txt | cut -c1-140
