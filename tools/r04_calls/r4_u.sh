#!/bin/bash
# GsrBinning.seg_len (entries per forward checkpoint / backward work item) 256 / 128 / 64: parity, then K6 / K7 kernel times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sl in 64 128; do
  echo "== parity with GSR_SEG_LEN=$sl"
  GSR_SEG_LEN=$sl timeout 900 python -m pytest $ROOT/tests/test_gpu_parity.py $ROOT/tests/test_views.py $ROOT/tests/test_fuzz.py $ROOT/tests/test_graph.py $ROOT/tests/test_score_views.py -m gpu -x -q 2>&1 | tail -3
  GSR_SEG_LEN=$sl timeout 900 python -m pytest $ROOT/tests/test_full_size.py -m gpu -x -q -k "C3 or needles or C2" 2>&1 | tail -3
done
for args in "--unbatched" "" "--gaussians 100000 --res 512 --unbatched" "--gaussians 100000 --res 512" "--init-opacity --unbatched"; do
  for sl in 256 128 64; do
    export GSR_SEG_LEN=$sl
    timeout 200 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --no-roofline --train-seconds 0 $args > $O/t.log 2>&1
    python $ROOT/tools/kstats.py $O/t 2>/dev/null | grep -E "k_render_bwd|k_render_fwd|k_work_order_bwd|steps" | sed "s/^/[$args] seg=$sl: /" | cut -c1-175
    rm -rf $O/t
  done
done
