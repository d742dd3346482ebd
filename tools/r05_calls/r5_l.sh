#!/bin/bash
# call L: forward-only calls write no checkpoints (GsrImages.ckpt NULL): the suite, the forward-only / score figures
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5l; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -12
timeout 300 python tools/bench_score.py 500000 1024 > $O/score_c3.json 2> $O/score.err; echo "score rc=$?"; cut -c1-700 $O/score_c3.json
timeout 300 python tools/bench_dropin.py --gaussians 500000 --res 1024 --seconds 0.8 --graphs 0 --streams 0,2 --patterns fb4,fwd > $O/dropin_c3.txt 2>&1; grep -E '^\{' $O/dropin_c3.txt | cut -c1-200
