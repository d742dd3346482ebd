#!/bin/bash
# call AM: the one-off widened fuzz tools on the final tree (after the row-message dwordx4 build and K8's zero_outside)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6am; mkdir -p $O; cd $ROOT
( timeout 1500 python tools/fuzz_views.py 300 0 2>&1 | tail -6 ) | tee $O/fuzz_views.txt
( timeout 1500 python tools/fuzz_big.py 12 100 2>&1 | tail -4 ) | tee $O/fuzz_big.txt
( timeout 900 python tools/fuzz_dropin.py 120 0 2>&1 | tail -4 ) | tee $O/fuzz_dropin.txt
( timeout 600 python tools/fuzz_rowmsg.py 300 1000 2>&1 | tail -2 ) | tee $O/fuzz_rowmsg.txt
( timeout 600 python tools/determinism_probe.py 6 2>&1 | tail -14 ) | tee $O/determinism.txt
