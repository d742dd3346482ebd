#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
BENCH_ARGS="--train-seconds 0" bash tools/kernel_times.sh r4i k7e1 k7e2 2>&1 | grep -E "render_bwd|preprocess_bwd"
