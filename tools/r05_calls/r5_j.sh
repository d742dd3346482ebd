#!/bin/bash
# call J: the whole -m gpu suite with the internal streams forced ON for every module built without a context
# (GSR_SIDE_STREAMS=2): every parity / reproducibility test of the per-view interface then runs its forwards on internal streams
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5j; mkdir -p $O; cd $ROOT
GSR_SIDE_STREAMS=2 timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 </dev/null > $O/pytest_streams.log 2>&1; echo "pytest(streams) rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_streams.log | tail -15
timeout 600 python -m pytest tests/test_graph.py -m gpu -q </dev/null > $O/pytest_graph.log 2>&1; echo "pytest(graph) rc=$?"; tail -2 $O/pytest_graph.log
