#!/bin/bash
# call AA: where K6's non-VALU quarter goes -- probes (timing only): the background tiles not written at all; the empty tiles at the
# HEAD of the work list instead of the tail. Kernel trace of the 4-view step each. (The two variants were built from render.hip
# with a `return` in front of the background store of render_fwd_body / bucket 0 first in k_work_order_fwd; the probes are not
# in the tree.)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5aa; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  if [ "$1" = base ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$1.so; fi
  timeout 150 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --no-roofline --train-seconds 0 $2 > $O/t.log 2>&1
  python $ROOT/tools/kstats.py $O/t 2>/dev/null | grep -E "k_render_fwd<false, 256>|k_render_bwd<256>|k_work_order_fwd" | sed "s/^/[$1 $2] /" | cut -c1-160
  grep -h '"metric"' $O/t.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   value', d['value'], d['ms_per_step'])"
  rm -rf $O/t
}
for r in 1 2; do for v in base skipempty emptyfirst; do run $v "--capture off"; done; done
for v in base skipempty emptyfirst; do run $v "--capture off --gaussians 100000 --res 512"; done
