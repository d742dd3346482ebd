#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py; raw outputs go to
# gpurun_out/<tag>/ (scratch). tools/profile_digest.py turns them into the committed summaries under profiles/.
# usage: tools/profile_round.sh <tag> [bench args...]      (every pass under its own timeout: nothing may stall the box)
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0 $*"
T="timeout 420"
$T rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- $BENCH --no-roofline > $OUT/pmc_fetch.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- $BENCH --no-roofline > $OUT/pmc_write.log 2>&1
$T rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/pmc_sq -o pmc -- $BENCH --no-roofline > $OUT/pmc_sq.log 2>&1
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line.json
ls -R $OUT | head -30
