#!/bin/bash
# call F: K1 (k_preprocess_views) at 5 / 6 waves per SIMD (96 / 80 VGPRs with 56 / 120 B of scratch per lane) against 4 (109 VGPRs,
# no scratch); the suites touching the pair counts after the "early only for the single-view entry" change
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5f; mkdir -p $O; cd $ROOT
timeout 600 python -m pytest tests/test_early_count.py tests/test_views.py tests/test_graph.py tests/test_gpu_parity.py tests/test_score_views.py -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
run() {
  if [ "$1" = base ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$1.so; fi
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --no-roofline --train-seconds 0 $2 > $O/t.log 2>&1
  python $ROOT/tools/kstats.py $O/t 2>/dev/null | grep -E "k_preprocess_views|k_os_hist|steps" | sed "s/^/[$1 $2] /" | cut -c1-150
  rm -rf $O/t
}
for r in 1 2; do for v in base k1w5 k1w6; do run $v ""; done; done
for v in base k1w5 k1w6; do run $v "--gaussians 2000000 --res 512"; done
