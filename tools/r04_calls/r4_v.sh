#!/bin/bash
# seg_len 64 / 128 after the K7 staging fix: a bounded smoke first (abort on failure), parity, then kernel times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4v; mkdir -p $O
cd $ROOT
for sl in 64 128; do
  GSR_SEG_LEN=$sl timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok seg $sl')" 2>&1 | tail -1 | cut -c1-200
  if [ ${PIPESTATUS[0]} -ne 0 ]; then echo "ABORT: smoke failed with seg $sl"; exit 1; fi
done
GSR_SEG_LEN=64 timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 | cut -c1-300
GSR_SEG_LEN=128 timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 | cut -c1-300
GSR_SEG_LEN=128 timeout 100 python -m pytest tests/test_full_size.py -m gpu -x -q -k "C3 and not init" 2>&1 | tail -2 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
run() {  # seg, bench args
  export GSR_SEG_LEN=$1; shift
  timeout 40 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --no-roofline --train-seconds 0 "$@" > $O/t.log 2>&1
  python $ROOT/tools/kstats.py $O/t 2>/dev/null | grep -E "k_render_bwd|k_render_fwd|steps" | sed "s/^/[$*] seg=$GSR_SEG_LEN: /" | cut -c1-170
  rm -rf $O/t
}
run 128 --unbatched; run 64 --unbatched
run 128 --gaussians 100000 --res 512 --unbatched; run 64 --gaussians 100000 --res 512 --unbatched
run 128 --gaussians 100000 --res 512; run 64 --gaussians 100000 --res 512
run 128
run 128 --init-opacity --unbatched
