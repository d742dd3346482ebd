#!/bin/bash
# call O: the per-view interface with 64 / 128 / 256 entries per backward work item (GSR_SEG_LEN), C3 and 100 k @512^2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6o; mkdir -p $O; cd $ROOT
for cfg in "" "--gaussians 100000 --res 512"; do
for r in 1 2; do for sl in 64 128 256; do
  GSR_SEG_LEN=$sl timeout 300 python tools/bench_dropin.py --graphs 0 --streams 0 --patterns fb4,per $cfg > $O/d.log 2>$O/d.err
  echo "seg $sl run $r $cfg: $(grep -o '"pattern": "[a-z0-9_]*", "views_per_s": [0-9.]*' $O/d.log | tr '\n' ' ')"
done; done; done
