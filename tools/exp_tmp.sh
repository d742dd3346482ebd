cd $GRAFT_REPO_ROOT
for flags in "" "-DGSR_EXP_MODE=2" "-DGSR_EXP_MODE=0"; do
python - <<PY
from dreamscene_amd import build
build.build(force=True, extra_flags="$flags".split())
PY
echo "== [$flags]"
python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d['max_grad_err_vs_oracle']))"
done
