#!/bin/bash
# call B: the -m gpu suite under the per-tensor relative bars (1e-5 * max|ref|, no floor of 1): which tests fail, by how much
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6b; mkdir -p $O; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -s </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|\[scene fixture\]" $O/pytest.log | tail -60
