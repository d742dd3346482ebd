#!/bin/bash
# final call: suite + smoke, the rocprofv3 evidence of the final tree (C3, init, C2, indoor), default bench line with CPU legs, sweep
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6final; mkdir -p $O; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -8
timeout 180 python -c "import __graft_entry__ as g; g.smoke()" </dev/null > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
bash tools/profile_all.sh r06 all > $O/profile_all.log 2>&1; tail -3 $O/profile_all.log
cd $ROOT
( time timeout 900 python bench.py </dev/null > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; tail -3 $O/bench_time.txt
python - <<PY
import json
try:
    d = json.load(open("$O/bench_default.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s")}, d["roofline"]["frac"], d["cpu_baseline"]["value"])
except Exception as e: print("bench failed", e)
PY
bash tools/sweep.sh r06 > $O/sweep.log 2>&1; tail -12 $O/sweep.log
# exchange kernels of the final tree (device side + kernel trace)
cd $ROOT
timeout 600 python tools/bench_exchange_device.py > $O/exchange_device_c3.json 2> $O/ex.err; echo "exchange c3 rc=$?"
timeout 600 python tools/bench_exchange_device.py --res 800 --views 1 > $O/exchange_device_c4.json 2>> $O/ex.err; echo "exchange c4 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $ROOT/tools/bench_exchange_device.py > $O/trace.log 2>&1
python $ROOT/tools/kstats.py $O/trace 2>/dev/null | grep -E "k_msg|k_rows|k_sum|kernel " | head -12 | tee $O/exchange_kernel_stats.txt
rm -rf $O/trace
