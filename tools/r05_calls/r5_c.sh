#!/bin/bash
# call C: forward-only internal streams (backward on the caller's stream), the fp32 fixture tests, the exchange formats on
# device tensors with two ranks, device-side exchange costs, a kernel profile of the importance-score paths
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5c; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_side_streams.py tests/test_golden.py tests/test_early_count.py tests/test_multirank_gpu.py tests/test_dropin_graphs.py tests/test_context.py -m gpu -q --maxfail=10 </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|CPU glue" $O/pytest.log | tail -20
timeout 300 python tools/bench_dropin.py --gaussians 500000 --res 1024 --seconds 0.8 --graphs 0 --streams 0,2,3 --patterns fb4,one_bw,fwd > $O/dropin_c3.txt 2>&1; echo "dropin c3 rc=$?"; grep -E '^\{|^SIDE' $O/dropin_c3.txt | cut -c1-200
timeout 300 python tools/bench_dropin.py --gaussians 100000 --res 512 --seconds 0.8 --graphs 0 --streams 0,2,3 --patterns fb4,one_bw,fwd > $O/dropin_c2.txt 2>&1; echo "dropin c2 rc=$?"; grep -E '^\{|^SIDE' $O/dropin_c2.txt | cut -c1-200
timeout 200 python tools/bench_exchange_device.py > $O/exchange_device_c3.json 2> $O/exchange_device_c3.err; echo "exchange rc=$?"; python - <<PY
import json
try:
    d = json.load(open("$O/exchange_device_c3.json"))
    for k, v in d["by_degree"].items():
        print(k, v["rows"], v["row_frac"], v["device_side_total_us"])
        if k == "D3":
            for kk, vv in v.items():
                if isinstance(vv, dict) and "gpu_us" in vv: print("   ", kk[:60], vv)
except Exception as e:
    print("no exchange json", e); print(open("$O/exchange_device_c3.err").read()[-1500:])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/score_trace -o trace -- python $ROOT/tools/bench_score.py 500000 1024 > $O/score.log 2>&1; echo "score rc=$?"; grep '^{' $O/score.log | cut -c1-900
python $ROOT/tools/kstats.py $O/score_trace > $O/score_kernel_stats.txt 2>&1; rm -rf $O/score_trace; head -32 $O/score_kernel_stats.txt
