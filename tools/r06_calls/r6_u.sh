#!/bin/bash
# call U: widened fuzz of the row-message kernels (row widths 1 ... 200, 1 ... 16 messages, word-boundary row counts)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6u; mkdir -p $O; cd $ROOT
timeout 1500 python tools/fuzz_rowmsg.py 300 0 > $O/fuzz_rowmsg.txt 2>&1; echo "rc=$?"; tail -12 $O/fuzz_rowmsg.txt | cut -c1-400
