cd $GRAFT_REPO_ROOT
for a in "50 10" "50 5" "30 10" "100 5" "50 10 --no-roofline"; do
 set -- $a
 echo "steps $1 warmup $2 $3: $(python bench.py --steps $1 --warmup $2 $3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
