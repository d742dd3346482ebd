#!/bin/bash
# call X: bench.py's N = 8 control flow -- eight ranks sharing the one GPU over gloo (collectives staged through the host), every
# wire format in the probe, default workload shape at a reduced size; then the same with 4 ranks at the C3 size
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r5x; mkdir -p $O; cd $ROOT
export GSR_BENCH_BACKEND=gloo GSR_BENCH_SHARE_GPU=1
timeout 900 python bench.py --gpus 8 --steps 6 --warmup 3 --gaussians 100000 --res 512 --no-cpu-baseline --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0 --exchange-probe-steps 3 --exchange-candidates dense,direct,rows,sparse_rs > $O/b8.json 2> $O/b8.err; echo "8 ranks rc=$?"
python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/b8.json") if l.startswith("{")][-1])
    print(d["n_gpus"], d["value"], d["ms_per_step"], d["exchange"]["format"], {k: v.get("ms_per_step") for k, v in d["exchange"]["probe_ms_per_step"].items()}, d["rccl"])
    print(d["config"]["parallelism"])
except Exception as e:
    print("no line:", e); print(open("$O/b8.err").read()[-3000:])
PY
timeout 900 python bench.py --gpus 4 --steps 6 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --rotate-seconds 0 --train-seconds 0 --exchange-probe-steps 3 > $O/b4.json 2> $O/b4.err; echo "4 ranks rc=$?"
python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/b4.json") if l.startswith("{")][-1])
    print(d["n_gpus"], d["value"], d["ms_per_step"], d["exchange"]["format"], {k: v.get("ms_per_step") for k, v in d["exchange"]["probe_ms_per_step"].items()})
except Exception as e:
    print("no line:", e); print(open("$O/b4.err").read()[-3000:])
PY
