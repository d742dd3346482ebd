cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for cfg in "--gaussians 500000 --res 1024" "--scene indoor --gaussians 2000000 --res 1024" "--gaussians 1000000 --res 512" "--gaussians 100000 --res 512"; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline $cfg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:round(v) for k,v in d['roofline']['stage_us_warmup'].items()})"
done
