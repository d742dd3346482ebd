#!/bin/bash
# call I: k_msg_apply with 4 / 3 / 2 row-instructions per trip; then the whole -m gpu suite and the default bench line of this tree
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6i; mkdir -p $O; cd $ROOT
for v in base t3 t2; do
  if [ $v = base ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$v.so; fi
  for cfg in "" "--res 800 --views 1"; do
  timeout 600 python tools/bench_exchange_device.py $cfg > $O/ex_$v.json 2> $O/ex.err
  python - <<PY
import json
try:
    d = json.load(open("$O/ex_$v.json"))["row_messages"]
    print("$v $cfg", {k: (x["pack_us (one launch)"], x["apply_us (one launch, W messages, rank-ordered sums stored)"]) for k, x in d.items()})
except Exception as e: print("$v failed", e)
PY
  done
done
unset GSR_LIB
tools/gpu_suite.sh r6i_suite
