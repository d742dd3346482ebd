cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_scene.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_scene.py 2>&1 | tail -1
python tools/bench_scene.py --scene object --models 4 --per-model 125000 --K 16 2>&1 | tail -1
