#!/bin/bash
# call AR: the invariant behind zero_outside checked the slow way inside the tests (GradArena.verify_zero_outside)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 1200 python -m pytest tests/test_k8_sparse.py tests/test_multirank_gpu.py tests/test_full_size.py -x -q -m gpu -k "not float64 and not vs_oracle and not reproducible and not properties and not multi_view_sum" 2>&1 | tail -5
