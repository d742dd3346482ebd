#!/bin/bash
# K7 work items ordered by remaining tile depth (GSR_BWD_ORDER=1, later made the only order) against the round-3 order
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for args in "" "--unbatched" "--gaussians 100000 --res 512 --unbatched" "--init-opacity --unbatched" "--init-opacity"; do
  for m in 0 1; do
    export GSR_BWD_ORDER=$m
    timeout 200 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --no-roofline --train-seconds 0 $args > $O/t_$m.log 2>&1
    python $ROOT/tools/kstats.py $O/t 2>/dev/null | grep -E "k_render_bwd|k_work_order_bwd|steps" | sed "s/^/[$args] order=$m: /" | cut -c1-160
    rm -rf $O/t
  done
done
