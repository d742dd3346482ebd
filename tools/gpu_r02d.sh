#!/bin/bash
# round 2, call D: captured graphs (fixed), ranged sort (specialised), trace profile
mkdir -p gpurun_out/r02d
O=gpurun_out/r02d
timeout 600 python -m pytest tests/test_graph.py tests/test_full_size.py tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-capture > $O/bench_nocapture.json 2> $O/bench_nocapture.err; echo "rc=$?"; tail -3 $O/bench_nocapture.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench_captured.json 2> $O/bench_captured.err; echo "rc=$?"; tail -3 $O/bench_captured.err
timeout 300 python bench.py --no-cpu-baseline --gaussians 100000 --res 512 > $O/bench_c2.json 2> $O/bench_c2.err; echo "rc=$?"; tail -3 $O/bench_c2.err
timeout 300 python bench.py --no-cpu-baseline --init-opacity > $O/bench_init.json 2> $O/bench_init.err; echo "rc=$?"; tail -3 $O/bench_init.err
timeout 300 python bench.py --no-cpu-baseline --views-per-step 1 > $O/bench_v1.json 2> $O/bench_v1.err; echo "rc=$?"; tail -3 $O/bench_v1.err
for f in bench_nocapture bench_captured bench_c2 bench_init bench_v1; do python - <<PY
import json
try:
    d=json.load(open("$O/$f.json"))
    print("$f", d["value"], "ms/step", d["ms_per_step"], "dropin", d["dropin_views_per_s"], "enq", d["host_enqueue_ms_per_step"], "wait", d["host_wait_ms_per_step"], d["config"].get("capture_stats"), d.get("exchange"))
    print("   ", d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["stage_us_per_view"])
except Exception as e: print("$f", e)
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py $O/trace 2>/dev/null | head -40
