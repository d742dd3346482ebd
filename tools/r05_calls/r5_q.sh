#!/bin/bash
# call Q: the default bench line of the final tree (call P's `/usr/bin/time` does not exist on the box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5q; mkdir -p $O; cd $ROOT
T0=$(date +%s)
timeout 600 python bench.py </dev/null > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s", "sustained_views_per_s")}, d["dropin_internal_streams"]["views_per_s"], d["rotating_cameras"]["views_per_s"], d["training_like"]["views_per_s"])
print(d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["traffic"], d["roofline"].get("traffic_kernel"), d["cpu_baseline"]["value"], d["max_grad_err_vs_oracle"].get("batched_sum"))
print(d["forward_only"])
PY
