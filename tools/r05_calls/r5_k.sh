#!/bin/bash
# call K: C3 on a crop against the independent float64 autograd oracle
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5k; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_full_size.py -m gpu -q -s -k "c3_window or c2_vs" </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "float64 autograd\]|passed|failed|Error|assert" $O/pytest.log | tail -30
free -g | head -2
