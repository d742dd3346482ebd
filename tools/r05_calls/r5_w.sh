#!/bin/bash
# call W: the importance-score paths under random configurations (tools/fuzz_score.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5w; mkdir -p $O; cd $ROOT
timeout 1500 python tools/fuzz_score.py ${1:-100} 0 > $O/fuzz_score.log 2>&1; echo "fuzz_score rc=$?"; tail -30 $O/fuzz_score.log | cut -c1-400
