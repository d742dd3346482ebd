#!/bin/bash
# call AA: bench line with the rotating / training-like legs timing both the eager and the captured path
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r6aa; mkdir -p $O; cd $ROOT
( time timeout 900 python bench.py --no-cpu-baseline </dev/null > $O/bench.json 2> $O/bench.err ) 2> $O/time.txt; echo "bench rc=$?"; tail -3 $O/time.txt; tail -3 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print(d["value"], d["config"]["capture_probe"], d["config"]["batched_through"])
print("rotating", json.dumps(d["rotating_cameras"])[:700])
print("training_like", json.dumps({k: v for k, v in d["training_like"].items() if k != "what"}))
print("init", json.dumps(d["init_state"]))
print(json.dumps(d["trainer_step"]["c3"]))
PY
