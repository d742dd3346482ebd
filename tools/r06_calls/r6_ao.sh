#!/bin/bash
# call AO: row-message fuzz with rows of up to 1024 floats (more than 64 dwordx4 chunks per row: the grouped walk)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 900 python tools/fuzz_rowmsg.py 400 5000 2>&1 | tail -6
