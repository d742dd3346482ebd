#!/bin/bash
# end of round 2: the whole -m gpu suite, smoke(), the remaining profile sets, the default bench line
O=gpurun_out/r02t; mkdir -p $O
S=$GRAFT_REPO_ROOT/gpurun_out/r02_summary; mkdir -p $S
timeout 420 python -m pytest tests -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 150 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" </dev/null > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
cp profiles/traffic.json $S/traffic.json
prof() {  # tag, bench args...
  tag=$1; shift
  timeout 300 bash tools/profile_round.sh $tag "$@" > $O/profile_$tag.log 2>&1 </dev/null
  GSR_PROFILE_OUT=$S timeout 120 python tools/profile_digest.py $tag > $O/digest_$tag.log 2>&1 </dev/null; tail -1 $O/digest_$tag.log | cut -c1-200
  cp gpurun_out/$tag/bench_line.json $S/${tag}_bench_line.json 2>/dev/null
  rm -rf gpurun_out/$tag
}
prof r02_init --init-opacity --no-dropin --capture off
prof r02_c2 --gaussians 100000 --res 512 --no-dropin --capture on
prof r02_indoor --scene indoor --gaussians 2000000 --no-dropin --capture off
timeout 420 python bench.py </dev/null > $S/r02_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 700 $S/r02_bench_default.json
du -sh gpurun_out
