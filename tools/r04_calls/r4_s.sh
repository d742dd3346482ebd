#!/bin/bash
# K7 launched over a fraction of the item capacity, the rest walked with a stride (GSR_BWD_GRID_DIV, removed): are the empty
# workgroups behind the real items what a single-view K7 pays for? (No.)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for args in "--unbatched" "" "--gaussians 100000 --res 512 --unbatched" "--init-opacity --unbatched"; do
  for m in "1 0" "4 0" "8 0" "16 0" "8 1"; do
    set -- $m
    export GSR_BWD_GRID_DIV=$1 GSR_BWD_ORDER=$2
    timeout 200 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --no-roofline --train-seconds 0 $args > $O/t.log 2>&1
    python $ROOT/tools/kstats.py $O/t 2>/dev/null | grep -E "k_render_bwd|steps" | sed "s/^/[$args] div=$1 order=$2: /" | cut -c1-170
    rm -rf $O/t
  done
done
