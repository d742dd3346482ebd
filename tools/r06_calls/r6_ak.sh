#!/bin/bash
# call AK: kernel traces of the rotating-cameras step through the eager and the captured path
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6ak; mkdir -p $O; cd $ROOT
for v in eager captured; do python tools/rotating_probe.py $v 300 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
for v in eager captured; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$v -o trace -- python $ROOT/tools/rotating_probe.py $v 100 > $O/trace_$v.log 2>&1
  echo "== $v"; tail -1 $O/trace_$v.log
  python $ROOT/tools/kstats.py $O/trace_$v 2>/dev/null | head -28 | cut -c1-130 | tee $O/kstats_$v.txt
  rm -rf $O/trace_$v
done
