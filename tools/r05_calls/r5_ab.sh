#!/bin/bash
# call AB: the whole -m gpu suite with the opt-ins forced on through the environment (what a user who exports them gets)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5ab; mkdir -p $O; cd $ROOT
GSR_SIDE_STREAMS=2 timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 </dev/null > $O/streams.log 2>&1; echo "GSR_SIDE_STREAMS=2 rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/streams.log | tail -8
GSR_DROPIN_GRAPHS=1 timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 </dev/null > $O/ring.log 2>&1; echo "GSR_DROPIN_GRAPHS=1 rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/ring.log | tail -8
GSR_SIDE_STREAMS=2 GSR_DROPIN_GRAPHS=1 timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 </dev/null > $O/both.log 2>&1; echo "both rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/both.log | tail -8
