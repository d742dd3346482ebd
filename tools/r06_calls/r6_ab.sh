#!/bin/bash
# call AB: host-side profile of the eager batched step (small scene: the GPU is not the pacer)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6ab; mkdir -p $O; cd $ROOT
timeout 600 python tools/host_profile.py 100000 512 > $O/host_profile_batched.txt 2>&1; echo "rc=$?"; head -75 $O/host_profile_batched.txt | cut -c1-150
