#!/bin/bash
# profile sets, digested on the box (raw databases are too big to copy back)
O=gpurun_out/r02h; mkdir -p $O
S=$GRAFT_REPO_ROOT/gpurun_out/r02_summary; mkdir -p $S
rm -f $S/traffic.json
prof() {  # tag, bench args...
  tag=$1; shift
  timeout 1500 bash tools/profile_round.sh $tag "$@" > $O/profile_$tag.log 2>&1
  GSR_PROFILE_OUT=$S python tools/profile_digest.py $tag > $O/digest_$tag.log 2>&1; tail -3 $O/digest_$tag.log
  cp gpurun_out/$tag/bench_line.json $S/${tag}_bench_line.json 2>/dev/null
  rm -rf gpurun_out/$tag
}
prof r02 --no-dropin --capture off
prof r02_init --init-opacity --no-dropin --capture off
prof r02_c2 --gaussians 100000 --res 512 --no-dropin --capture on
prof r02_dropin --unbatched
prof r02_indoor --scene indoor --gaussians 2000000 --no-dropin --capture off
cp $S/traffic.json profiles/traffic.json
timeout 500 python bench.py > $S/r02_bench_default.json 2> $O/bench_default.err; tail -c 600 $S/r02_bench_default.json; tail -3 $O/bench_default.err
du -sh gpurun_out
