cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_epilogue.py tests/test_scene.py -m gpu -x -q 2>&1 | tail -30
