cd $GRAFT_REPO_ROOT
for flags in "" "-DGSR_EXP_K1_DIRECT"; do
python - <<PY
from dreamscene_amd import build
build.build(force=True, extra_flags="$flags".split())
PY
echo "== [$flags]"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
for i in 1 2; do python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:round(v) for k,v in d['roofline']['stage_us_warmup'].items()})"; done
done
