#!/bin/bash
# identity last pass of the depth sort not copied: parity, kernel times (batched C3, single-view C3, indoor)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for args in "" "--unbatched" "--scene indoor --gaussians 2000000"; do
  echo "== $args"
  BENCH_ARGS="$args" bash tools/kernel_times.sh r4o 2>&1 | cut -c1-130
done
