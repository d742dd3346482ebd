#!/bin/bash
# call M: K1 without its colour (no SH row loaded or held) at 4 / 6 / 8 waves per SIMD: what a geometry pass alone would cost
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6m; mkdir -p $O; cd $ROOT
B="--no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0"
for r in 1 2; do for v in base k1p4 k1p6 k1p8; do
  if [ $v = base ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$v.so; fi
  timeout 300 python bench.py $B </dev/null > $O/o_$v$r.json 2>$O/o.err
  python - <<PY
import json
try:
    d=json.load(open("$O/o_$v$r.json")); print("$v $r", d["value"], d["roofline"]["stage_us_per_view"]["preprocess"])
except Exception as e: print("$v $r failed", e)
PY
done; done
