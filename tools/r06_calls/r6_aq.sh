#!/bin/bash
# call AQ: work-order kernels with their loads batched and kept: parity tests, kernel trace of the C3 step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6aq; mkdir -p $O; cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_views.py tests/test_graph.py tests/test_score_views.py -x -q -m gpu 2>&1 | tail -4
B="--no-cpu-baseline --no-dropin --sustain-seconds 0 --train-seconds 0 --rotate-seconds 0 --no-roofline --steps 60"
cd /tmp && export TMPDIR=/tmp
for r in 1 2; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $ROOT/bench.py $B > $O/trace.log 2>&1
tail -1 $O/trace.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], d['ms_per_step'])"
python $ROOT/tools/kstats.py $O/trace 2>/dev/null | grep -E "k_work_order|k_render_fwd<false, 256>|k_render_bwd<256>|GPU time per step" | head -6
rm -rf $O/trace
done
