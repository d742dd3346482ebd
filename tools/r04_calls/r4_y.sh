#!/bin/bash
# where the seg_len threshold sits: 2 and 8 views of C3, 4 views of 1 M / 200 k, with 128 and 256 forced (GPU time per step)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  export GSR_SEG_LEN=$1; shift
  timeout 40 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --no-roofline --train-seconds 0 "$@" > $O/t.log 2>&1
  python $ROOT/tools/kstats.py $O/t 2>/dev/null | grep -E "k_render_bwd|k_render_fwd|steps" | sed "s/^/[$*] seg=$GSR_SEG_LEN: /" | cut -c1-170
  rm -rf $O/t
}
for sl in 256 128; do run $sl --views-per-step 2; done
for sl in 256 128; do run $sl --gaussians 200000 --res 800; done
for sl in 256 128; do run $sl --gaussians 1000000 --res 512 --views-per-step 1 --unbatched; done
for sl in 256 128; do run $sl --gaussians 200000 --res 1024 --views-per-step 8; done
