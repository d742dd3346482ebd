#!/bin/bash
# K7: issue priority (s_setprio) for the waves with many candidates left; library variants built with -DGSR_K7_PRIO=<candidates
# per priority step> (removed)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for args in "--unbatched" "" "--gaussians 100000 --res 512 --unbatched" "--init-opacity --unbatched"; do
  echo "== $args"
  BENCH_ARGS="$args" bash tools/kernel_times.sh r4t prio16 prio32 prio48 2>&1 | grep -E "render_bwd|steps" | cut -c1-150
done
