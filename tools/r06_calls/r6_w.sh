#!/bin/bash
# call W: k_emit_cols with 1 / 4 / 8 runs (waves) per workgroup: binning parity subset, A/B, kernel trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6w; mkdir -p $O; cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_views.py -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -8
B="--no-cpu-baseline --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0"
for cfg in "" "--init-opacity --no-dropin" "--scene indoor --gaussians 2000000 --no-dropin"; do
for r in 1 2; do for v in new e1 e8; do
  if [ $v = new ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$v.so; fi
  timeout 300 python bench.py $B $cfg </dev/null > $O/o.json 2>$O/o.err
  python - <<PY
import json
try:
    d=json.load(open("$O/o.json")); print("$v $r [$cfg]", d["value"], d.get("dropin_views_per_s"), d["roofline"]["stage_us_per_view"]["duplicate"])
except Exception as e: print("$v $r failed", e)
PY
done; done; done
unset GSR_LIB
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0 > $O/trace.log 2>&1
python $ROOT/tools/kstats.py $O/trace > $O/kernel_stats.txt 2>&1; rm -rf $O/trace; grep -E "k_emit|k_row|k_col|kernel " $O/kernel_stats.txt | cut -c1-130; tail -1 $O/kernel_stats.txt
