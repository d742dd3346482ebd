cd $GRAFT_REPO_ROOT
for a in "" "--unbatched"; do
python bench.py --no-cpu-baseline --no-roofline --steps 200 --warmup 20 --gaussians 2000 --res 64 $a 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'])"
done
