cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_views.py -m gpu -x -q 2>&1 | tail -12
