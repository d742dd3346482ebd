#!/bin/bash
# call V: two-rank GPU tests with reduce_step (statistics beside the exchange)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6v; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_multirank_gpu.py tests/test_exchange_rows.py -m gpu -q </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error" $O/pytest.log | tail -12
