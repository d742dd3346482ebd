#!/bin/bash
# call J: slice messages (sparse reduce-scatter on the device): unit tests, two-rank tests, device-side timing at C3 / C4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6j; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_exchange_rows.py tests/test_multirank_gpu.py -m gpu -q </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -12
timeout 600 python tools/bench_exchange_device.py > $O/exchange_device_c3.json 2> $O/ex.err; echo "exchange c3 rc=$?"; tail -3 $O/ex.err
timeout 600 python tools/bench_exchange_device.py --res 800 --views 1 > $O/exchange_device_c4.json 2>> $O/ex.err; echo "exchange c4 rc=$?"
python - <<PY
import json
for n in ("c3", "c4"):
    try:
        d = json.load(open("$O/exchange_device_%s.json" % n))
        print(n, json.dumps(d["row_messages"]))
    except Exception as e: print(n, "failed", e)
PY
