#!/bin/bash
# round 2, call C: ranged sort + captured graphs + the rest; every command under its own timeout
mkdir -p gpurun_out/r02c
O=gpurun_out/r02c
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-capture > $O/bench_nocapture.json 2> $O/bench_nocapture.err; echo "rc=$?"; tail -c 1800 $O/bench_nocapture.json; tail -3 $O/bench_nocapture.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench_captured.json 2> $O/bench_captured.err; echo "rc=$?"; tail -c 1800 $O/bench_captured.json; tail -3 $O/bench_captured.err
timeout 300 python bench.py --no-cpu-baseline --gaussians 100000 --res 512 > $O/bench_c2.json 2> $O/bench_c2.err; echo "rc=$?"; tail -c 900 $O/bench_c2.json; tail -3 $O/bench_c2.err
timeout 420 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "rc=$?"; tail -c 2500 $O/bench_full.json; tail -3 $O/bench_full.err
