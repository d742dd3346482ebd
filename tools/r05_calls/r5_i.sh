#!/bin/bash
# call I: wide column-path scans (1 024 threads) + wave-aggregated bucket atomics in k_work_order_fwd against the previous library
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5i; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_fuzz.py tests/test_views.py tests/test_graph.py tests/test_score_views.py -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
run() {
  if [ "$1" = new ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$1.so; fi
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --no-roofline --train-seconds 0 $2 > $O/t.log 2>&1
  python $ROOT/tools/kstats.py $O/t 2>/dev/null | grep -E "k_radix_scan|k_work_order|steps" | sed "s/^/[$1 $2] /" | cut -c1-150
  rm -rf $O/t
}
for r in 1 2; do for v in new prev; do run $v ""; done; done
for v in new prev; do run $v "--unbatched"; done
for v in new prev; do run $v "--gaussians 100000 --res 512"; done
