#!/bin/bash
# call F: row messages second build (tests + device timing); round 4's tree against today's on the SAME box (K1 / K8: regression or
# box?); work-list classes per octave 1 / 4 / 8 (K6)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6f; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_exchange_rows.py tests/test_multirank_gpu.py -m gpu -q </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -12
timeout 600 python tools/bench_exchange_device.py > $O/exchange_device_c3.json 2> $O/ex.err; echo "exchange c3 rc=$?"
timeout 600 python tools/bench_exchange_device.py --res 800 --views 1 > $O/exchange_device_c4.json 2>> $O/ex.err; echo "exchange c4 rc=$?"
python - <<PY
import json
for n in ("c3", "c4"):
    try:
        d = json.load(open("$O/exchange_device_%s.json" % n))
        print(n, json.dumps(d["row_messages"]))
    except Exception as e: print(n, "failed", e)
PY
B="--no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0"
for r in 1 2; do
  for v in now r4; do
    if [ $v = r4 ]; then cd $ROOT/gpurun_scratch/r4; else cd $ROOT; fi
    timeout 300 python bench.py $B </dev/null > $O/k_$v$r.json 2>$O/k.err
    python - <<PY
import json
try:
    d=json.load(open("$O/k_$v$r.json")); print("$v $r", d["value"], d["roofline"]["stage_us_per_view"])
except Exception as e: print("$v $r failed", e)
PY
  done
done
cd $ROOT
for r in 1 2; do for v in base frac2 frac3; do
  if [ $v = base ]; then unset GSR_LIB; else export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_$v.so; fi
  for cfg in "" "--init-opacity"; do
  timeout 300 python bench.py $B $cfg </dev/null > $O/o_$v$r.json 2>$O/o.err
  python - <<PY
import json
try:
    d=json.load(open("$O/o_$v$r.json")); print("$v $r $cfg", d["value"], d["roofline"]["stage_us_per_view"]["render_fwd"], d["roofline"]["stage_us_per_view"]["render_bwd"])
except Exception as e: print("$v $r failed", e)
PY
  done
done; done
