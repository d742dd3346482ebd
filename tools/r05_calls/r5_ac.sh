#!/bin/bash
# call AC: the two test files that assumed the captured ring off, with GSR_DROPIN_GRAPHS=1 exported and without
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5ac; mkdir -p $O; cd $ROOT
GSR_DROPIN_GRAPHS=1 GSR_SIDE_STREAMS=2 timeout 600 python -m pytest tests/test_views.py tests/test_side_streams.py -m gpu -q </dev/null > $O/on.log 2>&1; echo "opt-ins on rc=$?"; tail -2 $O/on.log
timeout 600 python -m pytest tests/test_views.py tests/test_side_streams.py -m gpu -q </dev/null > $O/off.log 2>&1; echo "default rc=$?"; tail -2 $O/off.log
