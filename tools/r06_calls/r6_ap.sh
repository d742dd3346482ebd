#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 900 python -m pytest tests/test_full_size.py tests/test_exchange_rows.py -x -q -m gpu -k "changing_cameras or any_width" 2>&1 | grep -v "^$" | tail -40
