#!/bin/bash
# K1 with its loads hoisted (GSR_K1_HOIST, preprocess.hip) against the same tree without: parity first, then kernel times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_views.py tests/test_fuzz.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python -m pytest tests/test_full_size.py -m gpu -x -q -k "C3 or C2 or indoor or C5" 2>&1 | tail -3
for args in "" "--unbatched" "--gaussians 100000 --res 512" "--scene indoor"; do
  echo "== $args"
  BENCH_ARGS="$args" bash tools/kernel_times.sh r4q nohoist 2>&1 | grep -E "k_preprocess|render_bwd|steps" | cut -c1-150
done
