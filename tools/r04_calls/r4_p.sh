#!/bin/bash
# k_os_hist workgroup size (keys per workgroup): kernel times, batched and single view
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for args in "" "--unbatched" "--gaussians 100000 --res 512"; do
  echo "== $args"
  BENCH_ARGS="$args" bash tools/kernel_times.sh r4p hold h512t 2>&1 | grep -E "k_os_hist|steps" | cut -c1-130
done
