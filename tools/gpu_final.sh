#!/bin/bash
# the committed tree once more: whole -m gpu suite, smoke(), one default-shaped bench line without the CPU legs
O=gpurun_out/r02_final; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" </dev/null > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 120 python bench.py </dev/null --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
