#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
python tools/rotating_probe.py both 300 2>&1 | tail -5
python tools/rotating_probe.py captured 300 2>&1 | tail -2
