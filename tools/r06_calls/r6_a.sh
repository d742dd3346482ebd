#!/bin/bash
# call A: the atomic-footprint probe for a 4-entry K7 + the suite / bench of the tree as round 5 left it (baseline of this box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6a; mkdir -p $O; cd $ROOT
timeout 300 tools/probe/atomic_rows.bin > $O/atomic_rows.txt 2>&1; echo "probe rc=$?"; cat $O/atomic_rows.txt
tools/gpu_suite.sh r6a_suite
