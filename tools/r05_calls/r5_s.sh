#!/bin/bash
# call S: tools/fuzz_views.py again (the tool compared the arena's scales slot where per-view scales return through autograd)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5s; mkdir -p $O; cd $ROOT
timeout 1500 python tools/fuzz_views.py 600 0 > $O/fuzz_views.log 2>&1; echo "fuzz_views rc=$?"; tail -25 $O/fuzz_views.log
