#!/bin/bash
# round 2, call B: full GPU suite after the context refactor, seed-9 diagnostic, bench with the new fields, profile set r02
mkdir -p gpurun_out/r02b
python -m pytest tests -m gpu -q > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r02b/pytest.log
python tools/diag_fuzz_seed.py 9 3 0 > gpurun_out/r02b/diag.log 2>&1; cat gpurun_out/r02b/diag.log | tail -30
python bench.py > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err; tail -c 4000 gpurun_out/r02b/bench.json; tail -5 gpurun_out/r02b/bench.err
bash tools/profile_round.sh r02 > gpurun_out/r02b/profile.log 2>&1; tail -5 gpurun_out/r02b/profile.log
