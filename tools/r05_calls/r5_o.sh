#!/bin/bash
# call O: gsr_sum_slices (the local sum of the `direct` exchange in one pass) -- tests, device-side cost beside the torch adds
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5o; mkdir -p $O; cd $ROOT
timeout 600 python -m pytest tests/test_exchange_rows.py tests/test_multirank_gpu.py -m gpu -q </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 400 python tools/bench_exchange_device.py --gaussians 500000 --res 1024 > $O/exchange_device_c3.json 2> $O/ex.err; echo "exchange rc=$?"
python - <<PY
import json
d = json.load(open("$O/exchange_device_c3.json"))
for k, v in d.items():
    if isinstance(v, dict):
        for kk, vv in v.items():
            if isinstance(vv, dict):
                for k3, v3 in vv.items():
                    if "direct" in str(k3) or "total" in str(k3): print(k, kk, k3, v3)
            elif "direct" in str(kk) or "total" in str(kk): print(k, kk, vv)
PY
