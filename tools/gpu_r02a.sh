#!/bin/bash
# round 2, call A: parity with the deterministic gates + first timings
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q -s > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r02a/pytest.log
python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; tail -c 3000 gpurun_out/r02a/bench.json
python bench.py --init-opacity > gpurun_out/r02a/bench_init.json 2> gpurun_out/r02a/bench_init.err; tail -c 1500 gpurun_out/r02a/bench_init.json
python bench.py --unbatched --no-cpu-baseline > gpurun_out/r02a/bench_unbatched.json 2>&1; tail -c 1500 gpurun_out/r02a/bench_unbatched.json
