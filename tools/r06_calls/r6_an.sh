#!/bin/bash
# call AN: the V = 1 / per-view-scales hole fixed: its test, the suite's neighbours, fuzz_views again
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6an; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_k8_sparse.py tests/test_graph.py tests/test_views.py -x -q -m gpu 2>&1 | tail -8
( timeout 1500 python tools/fuzz_views.py 400 0 2>&1 | tail -8 ) | tee $O/fuzz_views.txt
