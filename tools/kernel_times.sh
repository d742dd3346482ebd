#!/bin/bash
# kernel-trace timing of K1 / K8 (and K7 as the yardstick of the box) for the product library and variants of it:
# usage: tools/kernel_times.sh <tag> [variant names: dreamscene_amd/libgsrast_<name>.so ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/$1; mkdir -p $O; shift
cd /tmp && export TMPDIR=/tmp
for v in new "$@"; do
  if [ $v = new ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$v.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/t_$v -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --no-roofline --train-seconds 0 $BENCH_ARGS > $O/t_$v.log 2>&1
  python $ROOT/tools/kstats.py $O/t_$v 2>/dev/null | grep -E "preprocess|render_bwd |steps" | sed "s/^/$v: /"
  rm -rf $O/t_$v
done
