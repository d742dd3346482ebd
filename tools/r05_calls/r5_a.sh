#!/bin/bash
# call A: the -m gpu suite on the round-5 tree (new: internal streams, timed-path parity, slot aliasing), the per-view interface
# under streams x graphs (tools/bench_dropin.py), the default bench line, the host profile of the per-view path
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5a; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -15
timeout 300 python tools/bench_dropin.py --gaussians 500000 --res 1024 --seconds 0.8 > $O/dropin_c3.txt 2>&1; echo "dropin c3 rc=$?"; cat $O/dropin_c3.txt | cut -c1-220
timeout 300 python tools/bench_dropin.py --gaussians 100000 --res 512 --seconds 0.8 > $O/dropin_c2.txt 2>&1; echo "dropin c2 rc=$?"; cat $O/dropin_c2.txt | cut -c1-220
timeout 400 python bench.py </dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s", "sustained_views_per_s")})
    r = d["roofline"]; print({k: r[k] for k in ("kernel", "frac", "traffic", "traffic_kernel", "avg_launch_us")}); print(r["whole_path"]); print(r["stage_us_per_view"])
    print(json.dumps(d["max_grad_err_vs_oracle"])[:1800])
except Exception as e:
    print("no bench line:", e)
PY
timeout 200 python tools/host_profile_dropin.py 100000 512 > $O/host_profile_dropin.txt 2>&1; head -40 $O/host_profile_dropin.txt
