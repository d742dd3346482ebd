#!/bin/bash
# call N: the binning chains of a batch forked into view groups on internal streams (api.hip ViewGroups, GSR_VIEW_GROUPS)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5n; mkdir -p $O; cd $ROOT
Q="--no-cpu-baseline --no-dropin --no-roofline --sustain-seconds 1.0 --rotate-seconds 0 --train-seconds 0"
for g in 1 2 4 1 2; do
  for cap in on off; do
    GSR_VIEW_GROUPS=$g timeout 200 python bench.py $Q --capture $cap > $O/b_g${g}_$cap.json 2> $O/b_g${g}_$cap.err
    python - $O/b_g${g}_$cap.json $g $cap <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"groups {sys.argv[2]} capture {sys.argv[3]}: value {d['value']:.0f} ms/step {d['ms_per_step']:.4f} sustained {d['sustained_views_per_s']:.0f}")
except Exception as e:
    print("groups", sys.argv[2], sys.argv[3], "failed", e)
PY
  done
done
GSR_VIEW_GROUPS=2 timeout 600 python -m pytest tests/test_views.py tests/test_graph.py tests/test_full_size.py -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for g in 1 2; do
  GSR_VIEW_GROUPS=$g timeout 200 python bench.py $Q --capture on --gaussians 100000 --res 512 > $O/s_g$g.json 2> $O/s_g$g.err
  python -c "
import json,sys
d=json.loads(open('$O/s_g$g.json').read().strip().splitlines()[-1]); print('100k@512 groups $g', d['value'], d['ms_per_step'])"
done
