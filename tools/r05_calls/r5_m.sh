#!/bin/bash
# call M: the exchange row kernels (gsr_rows_pack / gsr_rows_unpack): tests, two-rank formats on device tensors, device-side costs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5m; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_exchange_rows.py tests/test_multirank_gpu.py tests/test_abi.py -m gpu -q --maxfail=10 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error" $O/pytest.log | tail -15
timeout 300 python tools/bench_exchange_device.py > $O/exchange_device_c3.json 2> $O/exchange_device_c3.err; echo "exchange rc=$?"; tail -3 $O/exchange_device_c3.err
python - <<PY
import json
try:
    d = json.load(open("$O/exchange_device_c3.json"))
    for k, v in d["hip_kernels"].items():
        print("HIP", k, v["rows"], v["device_side_total_us"])
        for kk, vv in v.items():
            if isinstance(vv, dict) and "gpu_us" in vv: print("   ", kk[:70], vv)
    for k, v in d["torch_ops_by_degree"].items():
        print("torch", k, v["device_side_total_us"])
except Exception as e:
    print("no exchange json", e)
PY
