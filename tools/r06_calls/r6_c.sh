#!/bin/bash
# call C: binning without k_col_plan / k_work_order_fwd (both folded into their neighbours): suite, then A/B against round 5's library
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6c; mkdir -p $O; cd $ROOT
tools/ab_lib.sh r6c_ab libgsrast_r5.so
for r in 1 2; do for v in new old; do
  if [ $v = old ]; then export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_r5.so; else unset GSR_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0 </dev/null > $O/d_$v$r.json 2>$O/d.err
  python - <<PY
import json
try:
    d=json.load(open("$O/d_$v$r.json")); print("dropin $v $r", d["value"], d.get("dropin_views_per_s"), d.get("dropin_internal_streams"))
except Exception as e: print("$v $r failed", e)
PY
done; done
unset GSR_LIB
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0 > $O/trace.log 2>&1
python $ROOT/tools/kstats.py $O/trace > $O/kernel_stats.txt 2>&1; rm -rf $O/trace; head -32 $O/kernel_stats.txt
