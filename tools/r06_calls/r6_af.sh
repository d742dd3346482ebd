#!/bin/bash
# call AF: is K8 shorter with zero_outside? kernel traces of the captured step with and without
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6af; mkdir -p $O; cd $ROOT
B="--no-cpu-baseline --no-dropin --sustain-seconds 0 --train-seconds 0 --rotate-seconds 0 --no-roofline --steps 60"
cd /tmp && export TMPDIR=/tmp
for v in on off on off; do
  if [ $v = off ]; then export GSR_TMP_NO_ZO=1; else unset GSR_TMP_NO_ZO; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$v -o trace -- python $ROOT/bench.py $B > $O/trace_$v.log 2>&1
  echo "== $v"; tail -1 $O/trace_$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], d['config'].get('capture_stats'))"
  python $ROOT/tools/kstats.py $O/trace_$v 2>/dev/null | grep -E "k_preprocess_bwd|k_render_bwd|kernel " | head -4
  rm -rf $O/trace_$v
done
