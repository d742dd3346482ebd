#!/bin/bash
# call Z: camera-gradient sums of a workgroup in double: the six seeds against float64 again, the camera-gradient tests, seeds 0-400
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6z; mkdir -p $O; cd $ROOT
timeout 900 python tools/fuzz_seeds_vs_fp64.py 196 337 9 18 > $O/seeds_vs_fp64.txt 2>&1; echo "rc=$?"; grep -E "seed|dL_dview|dL_dproj|dL_dcampos" $O/seeds_vs_fp64.txt | cut -c1-160
GSR_FUZZ_SEEDS=24-400 timeout 2400 python -m pytest tests/test_fuzz.py tests/test_gpu_parity.py tests/test_context.py -m gpu -q </dev/null > $O/fuzz_seeds.log 2>&1; echo "fuzz seeds rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/fuzz_seeds.log | tail -12
grep -E "AssertionError: " $O/fuzz_seeds.log | sort | uniq -c | sort -rn | head -12 | cut -c1-200
