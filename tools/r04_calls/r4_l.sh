#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 600 python -m pytest tests/test_k8_sparse.py tests/test_scratch.py tests/test_views.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
BENCH_ARGS="" bash tools/kernel_times.sh r4l k8old 2>&1 | grep -E "preprocess_bwd|render_bwd |steps"
BENCH_ARGS="--init-opacity" bash tools/kernel_times.sh r4l k8old 2>&1 | grep -E "preprocess_bwd|steps"
