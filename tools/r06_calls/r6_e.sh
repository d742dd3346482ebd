#!/bin/bash
# call E: row messages (gsr_rowmsg_pack / _apply): unit tests, the two-rank GPU tests, device-side timing at C3 (4 views / rank) and
# C4 (800^2, 1 view / rank); the side-streams / host-logic tests after the per_view_accel change; default bench line incl. trainer_step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6e; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_exchange_rows.py tests/test_multirank_gpu.py tests/test_side_streams.py tests/test_dropin_graphs.py tests/test_abi.py -m gpu -q </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -12
timeout 600 python tools/bench_exchange_device.py > $O/exchange_device_c3.json 2> $O/ex.err; echo "exchange c3 rc=$?"; tail -2 $O/ex.err
timeout 600 python tools/bench_exchange_device.py --res 800 --views 1 > $O/exchange_device_c4.json 2>> $O/ex.err; echo "exchange c4 rc=$?"
python - <<PY
import json
for n in ("c3", "c4"):
    try:
        d = json.load(open("$O/exchange_device_%s.json" % n))
        print(n, json.dumps(d["row_messages"]))
        print(n, "before:", {k: v["device_side_total_us"] for k, v in d["hip_kernels"].items()})
    except Exception as e: print(n, "failed", e)
PY
timeout 600 python bench.py --no-cpu-baseline </dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json")); print(d["value"], d["dropin_views_per_s"]); print(json.dumps(d["trainer_step"])); print(json.dumps(d["max_grad_err_vs_oracle"])[:1500])
except Exception as e: print("bench failed", e)
PY
