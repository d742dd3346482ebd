#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5t; mkdir -p $O; cd $ROOT
timeout 300 python tools/r05_calls/r5_t.py 277 > $O/t.log 2>&1; echo "rc=$?"; tail -40 $O/t.log
