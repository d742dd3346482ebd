#!/bin/bash
# K7<128> with issue priority for the waves with many candidates left (GSR_K7_PRIO = candidates per priority step), small launches
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # lib, bench args
  if [ $1 = new ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$1.so; fi
  v=$1; shift
  timeout 40 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --no-roofline --train-seconds 0 "$@" > $O/t.log 2>&1
  python $ROOT/tools/kstats.py $O/t 2>/dev/null | grep -E "k_render_bwd|steps" | sed "s/^/[$*] $v: /" | cut -c1-170
  rm -rf $O/t
}
for v in new prio24 prio12; do run $v --unbatched; done
for v in new prio24 prio12; do run $v --gaussians 100000 --res 512 --unbatched; done
for v in new prio24; do run $v --gaussians 100000 --res 512; done
for v in new prio24; do run $v --init-opacity --unbatched; done
