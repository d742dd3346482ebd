#!/bin/bash
# early publishing of the one-sweep aggregates: parity, then kernel times A/B (batched C3, single-view C3, 100k)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_knn.py tests/test_fuzz.py tests/test_views.py -m gpu -q -x 2>&1 | tail -3
for args in "" "--unbatched" "--gaussians 100000 --res 512"; do
  echo "== $args"
  BENCH_ARGS="$args" bash tools/kernel_times.sh r4n noearly 2>&1 | grep -E "k_os_pass|k_os_hist|steps" | sed 's/us\/step/ /' | cut -c1-130
done
