cd $GRAFT_REPO_ROOT
for flags in "" "-DGSR_EXP_PRIO"; do
python - <<PY
from dreamscene_amd import build
build.build(force=True, extra_flags="$flags".split())
PY
echo "== [$flags]"
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['stage_us_warmup'])"; done
done
