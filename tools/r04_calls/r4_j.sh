#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4j; mkdir -p $O; cd $ROOT
timeout 300 python tools/bench_score.py > $O/score.json 2> $O/score.err; cat $O/score.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $ROOT/tools/bench_score.py > $O/trace.log 2>&1
python $ROOT/tools/kstats.py $O/trace 2>/dev/null | head -30 | cut -c1-150
