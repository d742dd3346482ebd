#!/bin/bash
# call AS: fresh seed ranges of the widened fuzz tools on the final tree
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6as; mkdir -p $O; cd $ROOT
( timeout 1500 python tools/fuzz_views.py 400 1000 2>&1 | tail -8 ) | tee $O/fuzz_views.txt
( timeout 900 python tools/fuzz_dropin.py 120 200 2>&1 | tail -3 ) | tee $O/fuzz_dropin.txt
( timeout 900 python tools/fuzz_score.py 2>&1 | tail -3 ) | tee $O/fuzz_score.txt
( timeout 1500 python tools/fuzz_big.py 12 200 2>&1 | tail -2 ) | tee $O/fuzz_big.txt
