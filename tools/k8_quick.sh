#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/$1; mkdir -p $O; cd $ROOT
GSR_LIB=$ROOT/dreamscene_amd/libgsrast_stamps.so python tools/k8_stamps.py 2>/dev/null | tail -12
timeout 900 python -m pytest tests/test_k8_sparse.py tests/test_scratch.py tests/test_views.py tests/test_fuzz.py tests/test_full_size.py tests/test_scene.py tests/test_epilogue.py tests/test_graph.py tests/test_multirank_gpu.py -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
tools/k8_variants.sh $1 k8old
BENCH_ARGS="--init-opacity" tools/k8_variants.sh $1 k8old
BENCH_ARGS="--gaussians 2000000 --res 512" tools/k8_variants.sh $1 k8old
