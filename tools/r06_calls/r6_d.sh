#!/bin/bash
# call D: only the forward work list folded into the ty pass (k_col_plan back): parity subset, A/B against round 5's library at C3
# (batched step and the per-view drop-in), and the trainer-shaped step of tools/train_step.py for the first time
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6d; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_full_size.py tests/test_gpu_parity.py tests/test_views.py tests/test_graph.py tests/test_early_count.py tests/test_scene.py tests/test_side_streams.py tests/test_dropin_graphs.py -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -8
for r in 1 2 3; do for v in new old; do
  if [ $v = old ]; then export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_r5.so; else unset GSR_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0 </dev/null > $O/d_$v$r.json 2>$O/d.err
  python - <<PY
import json
try:
    d=json.load(open("$O/d_$v$r.json")); print("$v $r", d["value"], d.get("dropin_views_per_s"), d["roofline"]["stage_us_per_view"])
except Exception as e: print("$v $r failed", e)
PY
done; done
unset GSR_LIB
timeout 600 python tools/bench_train_step.py --seconds 2 > $O/train_step.json 2> $O/train_step.err; echo "train_step rc=$?"; cat $O/train_step.json; tail -3 $O/train_step.err
timeout 600 python tools/bench_train_step.py --seconds 2 --init-opacity > $O/train_step_init.json 2>> $O/train_step.err; cat $O/train_step_init.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0 > $O/trace.log 2>&1
python $ROOT/tools/kstats.py $O/trace > $O/kernel_stats.txt 2>&1; rm -rf $O/trace; head -22 $O/kernel_stats.txt; tail -1 $O/kernel_stats.txt
