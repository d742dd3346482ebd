#!/bin/bash
# call H: K6 probes -- (a) the whole-tile forward variant (one pixel per lane, half the instructions per evaluation) forced on the
# 4-view C3 step; (b) K6's sensitivity to occupancy: 6 workgroups per CU (27 KB of LDS) against 3 / 2 (padded LDS)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  if [ "$1" = base ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$1.so; fi
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --no-roofline --train-seconds 0 $2 > $O/t.log 2>&1
  python $ROOT/tools/kstats.py $O/t 2>/dev/null | grep -E "k_render_fwd|k_render_bwd<256>|steps" | sed "s/^/[$1 $2] /" | cut -c1-150
  grep -h '"metric"' $O/t.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   value', d['value'], d['config']['batched_through'][:24])"
  rm -rf $O/t
}
for r in 1 2; do for v in base k6pad26000 k6pad52000; do run $v ""; done; done
run base "--fwd-mode 1"
run base "--fwd-mode 1 --init-opacity"
run base "--init-opacity"
