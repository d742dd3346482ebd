#!/bin/bash
# call AH: the init-state window against the independent float64 oracle
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6ah; mkdir -p $O; cd $ROOT
timeout 1200 python -m pytest tests/test_full_size.py -m gpu -q -s -k "float64" 2>&1 | grep -E "vs float64|passed|failed|Error|assert" | tee $O/fp64.log | tail -30
