#!/bin/bash
# call R: a one-off widening of the parity fuzz -- tests/test_fuzz.py on seeds 24..400 (single view vs the C oracle), random
# batched / captured configurations against the per-view calls (tools/fuzz_views.py), the exchange tests after the bounds check
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5r; mkdir -p $O; cd $ROOT
timeout 300 python -m pytest tests/test_exchange_rows.py -m gpu -q </dev/null > $O/exchange.log 2>&1; echo "exchange rc=$?"; tail -1 $O/exchange.log
GSR_FUZZ_SEEDS=24-400 timeout 1200 python -m pytest tests/test_fuzz.py -m gpu -q --maxfail=30 </dev/null > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/fuzz.log | tail -30
timeout 1500 python tools/fuzz_views.py 300 0 > $O/fuzz_views.log 2>&1; echo "fuzz_views rc=$?"; tail -25 $O/fuzz_views.log
