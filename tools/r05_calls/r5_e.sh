#!/bin/bash
# call E: the rocprofv3 evidence of round 5 -- C3 (trace + PMC passes), the per-view interface (trace; and with two internal
# streams: do the views' forwards overlap?), init / 100 k / indoor (trace), digested on the box into gpurun_out/r05_summary/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
export GSR_PROFILE_OUT=$ROOT/gpurun_out/r05_summary
mkdir -p $GSR_PROFILE_OUT
cp $ROOT/profiles/traffic.json $GSR_PROFILE_OUT/traffic.json 2>/dev/null
full() {  # <tag> <bench args...>: trace + 3 PMC passes
  local t=$1; shift
  bash $ROOT/tools/profile_round.sh $t "$@" > /dev/null 2>&1
  (cd $ROOT && python tools/profile_digest.py $t > $GSR_PROFILE_OUT/${t}_digest.log 2>&1)
  cp $ROOT/gpurun_out/$t/bench_line.json $GSR_PROFILE_OUT/${t}_bench_line.json 2>/dev/null
  rm -rf $ROOT/gpurun_out/$t/trace $ROOT/gpurun_out/$t/pmc_*
  echo "== $t"; head -22 $GSR_PROFILE_OUT/${t}_kernel_stats.txt | cut -c1-150
}
trace_only() {  # <tag> <bench args...>
  local t=$1; shift
  local O=$ROOT/gpurun_out/$t; mkdir -p $O
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0 "$@" > $O/trace.log 2>&1)
  grep -h '"metric"' $O/trace.log | tail -1 > $GSR_PROFILE_OUT/${t}_bench_line.json
  python tools/kstats.py $O/trace > $GSR_PROFILE_OUT/${t}_kernel_stats.txt 2>&1; rm -rf $O/trace
  echo "== $t"; head -16 $GSR_PROFILE_OUT/${t}_kernel_stats.txt | cut -c1-130
}
full r05 --no-dropin
trace_only r05_dropin --unbatched
trace_only r05_init --no-dropin --init-opacity
trace_only r05_c2 --no-dropin --gaussians 100000 --res 512
trace_only r05_indoor --no-dropin --scene indoor --gaussians 2000000
# two internal streams: the per-view interface, four forwards then four backwards
for ns in 0 2; do
  O=$ROOT/gpurun_out/r05_streams$ns; mkdir -p $O
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d $O/trace -o trace -- python $ROOT/tools/bench_dropin.py --graphs 0 --streams $ns --patterns fb4 --seconds 0.3 > $O/trace.log 2>&1)
  grep -h '^{' $O/trace.log > $GSR_PROFILE_OUT/r05_streams${ns}_overlap.txt
  python tools/overlap_digest.py $O/trace 0.5 72 >> $GSR_PROFILE_OUT/r05_streams${ns}_overlap.txt 2>&1; rm -rf $O/trace
  echo "== streams $ns"; head -8 $GSR_PROFILE_OUT/r05_streams${ns}_overlap.txt | cut -c1-220
done
ls $GSR_PROFILE_OUT
