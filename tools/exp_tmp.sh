cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_views.py -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-120
