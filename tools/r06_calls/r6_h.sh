#!/bin/bash
# call H: where k_msg_apply's 100 us go (variants without stores / without row loads); trainer-shaped step under the profiler
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6h; mkdir -p $O; cd $ROOT
for v in base p1 p2; do
  if [ $v = base ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$v.so; fi
  timeout 600 python tools/bench_exchange_device.py > $O/ex_$v.json 2> $O/ex.err
  python - <<PY
import json
try:
    d = json.load(open("$O/ex_$v.json"))["row_messages"]
    print("$v", {k: (x["pack_us (one launch)"], x["apply_us (one launch, W messages, rank-ordered sums stored)"]) for k, x in d.items()})
except Exception as e: print("$v failed", e)
PY
done
unset GSR_LIB
cd /tmp && export TMPDIR=/tmp
for leg in views_fused raw_leaves as_imported; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$leg -o trace -- python $ROOT/tools/bench_train_step.py --legs $leg --seconds 1 > $O/trace_$leg.log 2>&1
  python $ROOT/tools/kstats.py $O/trace_$leg > $O/train_${leg}_kernel_stats.txt 2>&1; rm -rf $O/trace_$leg
  echo "== $leg"; head -28 $O/train_${leg}_kernel_stats.txt | cut -c1-150; tail -1 $O/train_${leg}_kernel_stats.txt
done
