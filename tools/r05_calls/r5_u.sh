#!/bin/bash
# call U: random configurations between C2 and C3 in size against the C oracle (tools/fuzz_big.py); the views test added for means2D=None
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5u; mkdir -p $O; cd $ROOT
timeout 300 python -m pytest tests/test_views.py -m gpu -q </dev/null > $O/views.log 2>&1; echo "views rc=$?"; tail -1 $O/views.log
timeout 1500 python tools/fuzz_big.py 24 0 > $O/fuzz_big.log 2>&1; echo "fuzz_big rc=$?"; tail -30 $O/fuzz_big.log | cut -c1-260
