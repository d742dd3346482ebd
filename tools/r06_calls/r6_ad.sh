#!/bin/bash
# call AD: apply_slices with one message per word (MERGE = 1); rows apply with 1 / 2 dwordx4 row-instructions per trip
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6ad; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_exchange_rows.py tests/test_multirank_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/fuzz_rowmsg.py 200 0 2>&1 | tail -2
for v in base trip1 base trip1; do
  if [ $v = base ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$v.so; fi
  for c in c3 c4; do
    if [ $c = c3 ]; then A=""; else A="--res 800 --views 1"; fi
    timeout 600 python tools/bench_exchange_device.py $A > $O/ex_${v}_$c.json 2> $O/ex.err; 
    python - <<PY
import json
try:
    d = json.load(open("$O/ex_${v}_$c.json"))["row_messages"]
    for k, x in d.items():
        s = x["sparse_rs_device"]
        print("$v $c", k, "rows: pack", x["pack_us (one launch)"], "apply", x["apply_us (one launch, W messages, rank-ordered sums stored)"],
              "| sparse_rs: pack", s["pack_slices_us"], "reduce", s["reduce_owned_us"], "apply", s["apply_slices_us"])
except Exception as e: print("$v $c failed", e)
PY
  done
done
