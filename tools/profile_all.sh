#!/bin/bash
# One gpurun call: the rocprofv3 evidence of a round -- kernel trace + PMC passes of the default bench (C3) and of the named
# variants, digested ON THE BOX into gpurun_out/<tag>_summary/ (the raw databases are too big to travel back); copy that
# directory's files into profiles/ afterwards. usage: tools/profile_all.sh r03 [c3|all]
TAG=${1:-r03}; WHAT=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export GSR_PROFILE_OUT=$ROOT/gpurun_out/${TAG}_summary
mkdir -p $GSR_PROFILE_OUT
cp $ROOT/profiles/traffic.json $GSR_PROFILE_OUT/traffic.json 2>/dev/null   # the digests update its entries
run() {  # <tag> <bench args...>
  local t=$1; shift
  bash $ROOT/tools/profile_round.sh $t --no-dropin "$@" > /dev/null 2>&1
  (cd $ROOT && python tools/profile_digest.py $t > $GSR_PROFILE_OUT/${t}_digest.log 2>&1)
  cp $ROOT/gpurun_out/$t/bench_line.json $GSR_PROFILE_OUT/${t}_bench_line.json 2>/dev/null
  rm -rf $ROOT/gpurun_out/$t/trace $ROOT/gpurun_out/$t/pmc_*      # the raw databases stay on the box
  echo "== $t"; head -14 $GSR_PROFILE_OUT/${t}_kernel_stats.txt | cut -c1-150
}
run $TAG
if [ "$WHAT" = all ]; then
  run ${TAG}_init --init-opacity
  run ${TAG}_c2 --gaussians 100000 --res 512
  run ${TAG}_indoor --scene indoor --gaussians 2000000
fi
ls $GSR_PROFILE_OUT
