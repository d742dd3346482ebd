#!/bin/bash
# call V: the per-view interface under random configurations / call patterns with internal streams and / or the captured ring on
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5v; mkdir -p $O; cd $ROOT
timeout 1500 python tools/fuzz_dropin.py ${1:-120} 0 > $O/fuzz_dropin.log 2>&1; echo "fuzz_dropin rc=$?"; tail -30 $O/fuzz_dropin.log | cut -c1-400
