#!/bin/bash
# round 4, call e: folds (K1 clears the sort state; scan+plan; work list in the tail of the row pass): parity, then A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4e; mkdir -p $O; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -15
for r in 1 2; do
for v in new h8k; do
  if [ $v = new ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --rotate-seconds 0 --train-seconds 0 --sustain-seconds 1 </dev/null > $O/b_$v.json 2>$O/b_$v.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b_$v.json")); print("$v $r", d["value"], d["dropin_views_per_s"], d["roofline"]["stage_us_per_view"])
except Exception as e: print("$v $r failed", e)
PY
done
done
unset GSR_LIB
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --rotate-seconds 0 --train-seconds 0 --sustain-seconds 0 > $O/trace.log 2>&1
python $ROOT/tools/kstats.py $O/trace > $O/kernel_stats.txt 2>&1; head -24 $O/kernel_stats.txt | cut -c1-150
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_d -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --unbatched --rotate-seconds 0 --train-seconds 0 --sustain-seconds 0 > $O/trace_d.log 2>&1
python $ROOT/tools/kstats.py $O/trace_d > $O/kernel_stats_dropin.txt 2>&1; head -24 $O/kernel_stats_dropin.txt | cut -c1-150
