#!/bin/bash
# call P: the whole -m gpu suite, smoke() and a short bench of the tree after the last changes (auto exchange without host read)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
tools/gpu_suite.sh r6p_suite
