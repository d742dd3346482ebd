#!/bin/bash
# call X: the one-off widened fuzz of round 5 again, under round 6's per-tensor relative bars (seeds 24-400 of tests/test_fuzz.py;
# tools/fuzz_big.py; tools/fuzz_views.py; tools/fuzz_score.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6x; mkdir -p $O; cd $ROOT
GSR_FUZZ_SEEDS=24-400 timeout 2400 python -m pytest tests/test_fuzz.py -m gpu -q </dev/null > $O/fuzz_seeds.log 2>&1; echo "fuzz seeds rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/fuzz_seeds.log | tail -15
grep -E "AssertionError: " $O/fuzz_seeds.log | sort | uniq -c | sort -rn | head -20 | cut -c1-200
timeout 1500 python tools/fuzz_big.py > $O/fuzz_big.txt 2>&1; echo "fuzz_big rc=$?"; tail -4 $O/fuzz_big.txt | cut -c1-300
timeout 1500 python tools/fuzz_views.py > $O/fuzz_views.txt 2>&1; echo "fuzz_views rc=$?"; tail -3 $O/fuzz_views.txt | cut -c1-300
timeout 900 python tools/fuzz_score.py > $O/fuzz_score.txt 2>&1; echo "fuzz_score rc=$?"; tail -3 $O/fuzz_score.txt | cut -c1-300
