#!/bin/bash
# call Y: the six widened-fuzz seeds beyond the relative bar: HIP and the fp32 C oracle against float64 autograd
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6y; mkdir -p $O; cd $ROOT
timeout 900 python tools/fuzz_seeds_vs_fp64.py 107 196 243 260 337 381 > $O/seeds_vs_fp64.txt 2>&1; echo "rc=$?"; cat $O/seeds_vs_fp64.txt | cut -c1-200
