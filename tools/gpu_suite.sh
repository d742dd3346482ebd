#!/bin/bash
# Runs on the GPU box (gpurun): the -m gpu suite, smoke(), one bench line without the CPU legs, a kernel trace of it.
# usage: tools/gpu_suite.sh <tag> [pytest args]
TAG=${1:-suite}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 "$@" </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" </dev/null > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 300 python bench.py </dev/null --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s")}, d["roofline"]["stage_us_per_view"], d["roofline"]["avg_launch_us"])
except Exception as e:
    print("no bench line:", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0 > $O/trace.log 2>&1
python $ROOT/tools/kstats.py $O/trace > $O/kernel_stats.txt 2>&1; rm -rf $O/trace; head -40 $O/kernel_stats.txt
