#!/bin/bash
# call N: the rocprofv3 evidence of round 6 (kernel traces + PMC passes: C3, init state, C2, indoor), the default bench line with the
# CPU baseline legs, the drop-in trace, the sweep
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6n; mkdir -p $O; cd $ROOT
bash tools/profile_all.sh r06 all > $O/profile_all.log 2>&1; tail -5 $O/profile_all.log
cd $ROOT
( time timeout 900 python bench.py </dev/null > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; tail -3 $O/bench_time.txt
python - <<PY
import json
try:
    d = json.load(open("$O/bench_default.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "dropin_views_per_s")}, d["roofline"]["frac"], d["cpu_baseline"]["value"])
    print(json.dumps(d["max_grad_err_vs_oracle"])[:1200])
    print(json.dumps(d["trainer_step"])[:900])
except Exception as e: print("bench failed", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_dropin -o trace -- python $ROOT/tools/bench_dropin.py --graphs 0 --streams 0 --patterns fwd4_bwd4 --seconds 1.5 > $O/trace_dropin.log 2>&1
python $ROOT/tools/kstats.py $O/trace_dropin > $O/r06_dropin_kernel_stats.txt 2>&1; rm -rf $O/trace_dropin; head -26 $O/r06_dropin_kernel_stats.txt | cut -c1-140; tail -1 $O/r06_dropin_kernel_stats.txt
cd $ROOT
bash tools/sweep.sh r06 > $O/sweep.log 2>&1; cat $O/sweep.log
