#!/bin/bash
# call AC: row-message kernels with one dwordx4 per lane on the message side (fifth build): parity tests, fuzz, device-side timing
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6ac; mkdir -p $O; cd $ROOT
timeout 900 python -m pytest tests/test_exchange_rows.py tests/test_multirank_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python tools/fuzz_rowmsg.py 200 0 2>&1 | tail -4
timeout 600 python tools/bench_exchange_device.py > $O/exchange_device_c3.json 2> $O/ex.err; echo "exchange c3 rc=$?"; tail -3 $O/ex.err
timeout 600 python tools/bench_exchange_device.py --res 800 --views 1 > $O/exchange_device_c4.json 2>> $O/ex.err; echo "exchange c4 rc=$?"
python - <<PY
import json
for n in ("c3", "c4"):
    try:
        d = json.load(open("$O/exchange_device_%s.json" % n))["row_messages"]
        for k, x in d.items():
            s = x["sparse_rs_device"]
            print(n, k, "rows: pack", x["pack_us (one launch)"], "apply", x["apply_us (one launch, W messages, rank-ordered sums stored)"],
                  "| sparse_rs: pack", s["pack_slices_us"], "reduce", s["reduce_owned_us"], "apply", s["apply_slices_us"], "wire MB", s["wire_bytes_per_rank"] / 1e6)
    except Exception as e: print(n, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $ROOT/tools/bench_exchange_device.py > $O/trace.log 2>&1
python $ROOT/tools/kstats.py $O/trace 2>/dev/null | grep -E "k_msg|k_rows|k_sum|kernel " | head -12 | tee $O/exchange_kernel_stats.txt
rm -rf $O/trace
cd $ROOT
timeout 900 python -m pytest tests/test_views.py tests/test_graph.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json; python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("value", d["value"], "rot", d.get("rotating_cameras"), "tl", json.dumps(d.get("training_like"))[:400])
PY
