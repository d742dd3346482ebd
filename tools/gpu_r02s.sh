#!/bin/bash
# after the sparse K8: refreshed profile sets of the default workload (batched and drop-in), digested on the box
O=gpurun_out/r02s; mkdir -p $O
S=$GRAFT_REPO_ROOT/gpurun_out/r02_summary; mkdir -p $S
cp profiles/traffic.json $S/traffic.json
prof() {  # tag, bench args...
  tag=$1; shift
  timeout 400 bash tools/profile_round.sh $tag "$@" > $O/profile_$tag.log 2>&1 </dev/null
  GSR_PROFILE_OUT=$S timeout 120 python tools/profile_digest.py $tag > $O/digest_$tag.log 2>&1 </dev/null; tail -3 $O/digest_$tag.log
  cp gpurun_out/$tag/bench_line.json $S/${tag}_bench_line.json 2>/dev/null
  rm -rf gpurun_out/$tag
}
prof r02 --no-dropin --capture off
prof r02_dropin --unbatched
cp $S/traffic.json profiles/traffic.json
timeout 200 python bench.py --no-cpu-baseline > $S/r02_bench_default_nocpu.json 2> $O/bench_default.err </dev/null; tail -c 900 $S/r02_bench_default_nocpu.json; tail -3 $O/bench_default.err
du -sh gpurun_out
